R=$GRAFT_REPO_ROOT
timeout 600 python3 $R/tools/gpu_check.py infer_config1 infer_ragged vc_noise_control flow 2>&1 | tail -5
for sp in 1 2 4 8; do
HSP_FRONT_SPLITS=$sp timeout 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('splits=$sp', d['ms_per_step'])"
done
