#!/bin/bash
# plain low-channel convs: two-per-CU configs (debug 0) against the one-per-CU ones (debug 2048)
set -e
for dbg in 0 2048; do
  for k in 3 7 11; do python tools/conv_bench.py --cin 64 --cout 64 --k $k --len 32000 --act 0 --res 1 --debug $dbg --reps 20; done
  for k in 3 7 11; do python tools/conv_bench.py --cin 32 --cout 32 --k $k --len 64000 --act 0 --res 1 --debug $dbg --reps 20; done
done
python tools/conv_bench.py --cin 64 --cout 64 --k 11 --dil 5 --len 32000 --act 0 --res 1 --reps 20
python tools/conv_bench.py --cin 256 --cout 256 --k 7 --len 400 --act 0 --res 1 --reps 20
