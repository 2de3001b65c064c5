#!/bin/bash
# decomposition of a 128x128-tile launch with the kernel's tuning switches (hsp_conv1d_args.debug, results are then
# WRONG): full (0) / no epilogue (16) / no epilogue + producers stage chunk 0 only (17) / consumers skip MFMAs (2) /
# neither staging nor MFMAs (3) / scalar epilogue (32).  DESIGN.md §5 item 8.
set -e
for k in 3 11; do
  for dbg in 0 16 17 2 3 32; do
    python tools/conv_bench.py --cin 256 --cout 256 --k $k --len 4000 --act 0 --res 1 --debug $dbg --reps 20
  done
done
for dbg in 0 16 17; do python tools/conv_bench.py --cin 128 --cout 128 --k 3 --len 16000 --act 0 --res 1 --debug $dbg --reps 20; done
