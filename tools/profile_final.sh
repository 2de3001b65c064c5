# rocprofv3 summary of the bench command (eager launches on one stream: graph-replayed kernels are not
# visible to --kernel-trace, and concurrent streams would overlap the per-kernel durations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
HSP_AMP_STREAMS=0 HSP_FRONT_SPLITS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
grep '^{' $R/gpurun_out/prof_final.log | tail -1 > $R/gpurun_out/prof_final_bench.json
cd $R && timeout 280 python3 bench.py --dump-launches gpurun_out/launches_final.txt > gpurun_out/bench_final.log 2>&1; grep '^{' gpurun_out/bench_final.log | tail -1 > gpurun_out/bench_final.json; cut -c1-200 gpurun_out/bench_final.json
