# per-dispatch kernel trace of ONE eager Megatts2PLM1.infer call (B=16, T=200) -> gpurun_out/plm_trace/*kernel_trace.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/plm_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/plm_trace -- python3 $R/tools/plm_bench.py --no-graph --reps 1 ${PLM_ARGS} > $R/gpurun_out/plm_trace.log 2>&1
grep -E "eager|graph" $R/gpurun_out/plm_trace.log
find $R/gpurun_out/plm_trace -name "*.csv" | head
