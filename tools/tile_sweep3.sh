#!/bin/bash
# T = 200 launches of the flows: tile shape x batch-group size
set -e
for b in 8 32; do
  for dbg in 0 16384 256 8192 4096; do
    python tools/conv_bench.py --cin 192 --cout 768 --k 5 --len 200 --batch $b --act 0 --debug $dbg --reps 50
  done
done
