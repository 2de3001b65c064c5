# rocprofv3 kernel stats of the PLM loop (eager launches; graph-replayed kernels are invisible to --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/plm_prof -- python3 $R/tools/plm_bench.py --no-graph --reps 1 ${PLM_ARGS} > $R/gpurun_out/plm_prof.log 2>&1
grep -E "eager|graph" $R/gpurun_out/plm_prof.log
