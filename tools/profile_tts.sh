# Config-3 (full TTS, batch 16) measurements: whole-step JSON with per-stage split, and rocprofv3 kernel
# stats of the PLM loop (eager launches: graph-replayed kernels are invisible to --kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 300 python3 tools/tts_bench.py --steps 5 --warmup 2 > gpurun_out/tts_bench.log 2>&1; grep '^{' gpurun_out/tts_bench.log | tail -1 > gpurun_out/tts_bench.json
timeout 300 python3 tools/plm_bench.py --reps 5 > gpurun_out/plm_bench.log 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/plm_prof2 -- python3 $R/tools/plm_bench.py --no-graph --reps 1 > $R/gpurun_out/plm_prof2.log 2>&1
cd $R && cat gpurun_out/tts_bench.json | cut -c1-900; tail -3 gpurun_out/plm_bench.log
