# (round 6) the fused activation prologue of the direct conv against the stand-alone activation + plain conv at the
# low-channel stages (the HBM-bound ones): per-launch times.
cd $GRAFT_REPO_ROOT
for shp in "32 64000 3" "32 64000 7" "32 64000 11" "64 32000 3" "64 32000 7"; do
  set -- $shp
  python tools/conv_bench.py --cin $1 --cout $1 --k $3 --len $2 --act 0 --res 1 --reps 30
  python tools/conv_bench.py --cin $1 --cout $1 --k $3 --len $2 --act 1 --res 1 --reps 30
  python tools/conv_bench.py --cin $1 --cout $1 --k $3 --len $2 --actonly 1 --reps 30
done
