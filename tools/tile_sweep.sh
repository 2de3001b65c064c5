#!/bin/bash
# tile-shape comparison for the 512-channel stage (L = 800) and the SourceNetwork shapes
set -e
for dbg in 0 256 512; do
  for k in 3 7 11; do python tools/conv_bench.py --cin 512 --cout 512 --k $k --len 800 --act 0 --res 1 --debug $dbg --reps 20; done
done
for dbg in 0 256 1024; do
  python tools/conv_bench.py --cin 256 --cout 256 --k 7 --len 400 --act 0 --res 1 --debug $dbg --reps 20
  python tools/conv_bench.py --cin 256 --cout 256 --k 7 --len 4000 --act 0 --res 1 --debug $dbg --reps 20
done
