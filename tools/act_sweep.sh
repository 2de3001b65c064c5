#!/bin/bash
# stand-alone activation kernel at the vocoder's stage shapes (B = 32)
set -e
for cl in "512 800" "256 4000" "128 16000" "64 32000" "32 64000"; do
  set -- $cl
  python tools/conv_bench.py --actonly 1 --cin $1 --cout $1 --len $2 --reps 20
done
