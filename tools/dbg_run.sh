R=$GRAFT_REPO_ROOT
timeout 600 python3 $R/tools/gpu_check.py generator source_network infer_config1 amp_k7_c32 2>&1 | tail -5
for st in 0 1; do
HSP_AMP_STREAMS=$st timeout 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams=$st', d['ms_per_step'])"
done
HSP_AMP_STREAMS=1 timeout 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams=1 eager', d['ms_per_step'])"
