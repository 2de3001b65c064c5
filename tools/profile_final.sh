# rocprofv3 summary of the bench command (eager launches on one stream: graph-replayed kernels are not
# visible to --kernel-trace, and concurrent streams would overlap the per-kernel durations).  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export HSP_SERIAL_STREAMS=1   # the product step's own launches, one after the other on one stream
rm -rf $R/gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-extra > $R/gpurun_out/prof_final.log 2>&1
grep '^{' $R/gpurun_out/prof_final.log | tail -1 > $R/gpurun_out/prof_final_bench.json
find $R/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_final_kernel_stats.csv
head -12 $R/gpurun_out/prof_final_kernel_stats.csv
