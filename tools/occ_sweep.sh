#!/bin/bash
# main loop only (debug 17: no epilogue, producers stage chunk 0 only): one workgroup per CU (B = 4, 256 tiles of
# 128x128) against two (B = 8) -- the MFMA rate a lone workgroup reaches
set -e
for b in 4 8 16; do
  python tools/conv_bench.py --cin 256 --cout 256 --k 11 --len 4000 --batch $b --act 0 --res 1 --debug 17 --reps 30
done
for b in 4 8; do
  python tools/conv_bench.py --cin 256 --cout 256 --k 11 --len 4000 --batch $b --act 0 --res 1 --debug 0 --reps 30
done
