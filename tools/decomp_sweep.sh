#!/bin/bash
# decomposition of a 128x128-tile launch: full (0) / no epilogue (16) / no epilogue + producers stage chunk 0 only (17) / no residual
set -e
export HSP_LIB=$PWD/variants/libhsp_m128.so
S=131072
for k in 3 11; do
  for dbg in 0 16 17; do
    python tools/conv_bench.py --cin 256 --cout 256 --k $k --len 4000 --act 0 --res 1 --debug $((S+dbg)) --reps 20
  done
done
for dbg in 0 16 17; do python tools/conv_bench.py --cin 128 --cout 128 --k 3 --len 16000 --act 0 --res 1 --debug $((S+dbg)) --reps 20; done
python tools/conv_bench.py --cin 128 --cout 128 --k 3 --len 16000 --act 0 --res 0 --debug $S --reps 20
