# HBM traffic of one bench step from the PMC counters (separate passes, MI355X_MICROARCH.md "HBM").
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-roofline > $R/gpurun_out/traffic_$c.log 2>&1
done
# calibration: a pure streaming copy-like kernel with known bytes (stand-alone activation: 1 read + 1 write of 64 MB)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/traffic_cal_f -- python3 $R/tools/conv_bench.py --cin 128 --cout 128 --k 3 --len 16000 --batch 8 --act 0 --reps 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic_cal_w -- python3 $R/tools/conv_bench.py --cin 128 --cout 128 --k 3 --len 16000 --batch 8 --act 0 --reps 2 > /dev/null 2>&1
ls $R/gpurun_out | grep traffic
