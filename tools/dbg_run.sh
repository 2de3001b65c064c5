R=$GRAFT_REPO_ROOT
timeout 600 python3 $R/tools/gpu_check.py 2>&1 | grep -v PASS | tail -5
for fc in 0 32 64 128 512; do
HSP_FUSE_ACT_MAX_C=$fc timeout 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse<=$fc', d['ms_per_step'])"
done
