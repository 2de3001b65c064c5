# per-dispatch kernel trace of eager vocoder steps in the PRODUCT launch mix (AMP chains / front groups on streams)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/step_trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/step_trace -- python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-roofline > $R/gpurun_out/step_trace.log 2>&1
tail -2 $R/gpurun_out/step_trace.log | cut -c1-400
