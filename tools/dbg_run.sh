R=$GRAFT_REPO_ROOT
for fc in 0 32 64; do
HSP_FUSE_ACT_MAX_C=$fc timeout 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse<=$fc', d['ms_per_step'])"
done
python3 $R/tools/conv_bench.py --cin 64 --cout 64 --k 3 --len 32000 --batch 8 --act 0 --res 1 2>&1 | grep cin
python3 $R/tools/conv_bench.py --cin 32 --cout 32 --k 3 --len 64000 --batch 8 --act 0 --res 1 2>&1 | grep cin
python3 $R/tools/conv_bench.py --cin 32 --cout 32 --k 11 --len 64000 --batch 8 --act 0 --res 1 2>&1 | grep cin
python3 $R/tools/conv_bench.py --cin 32 --cout 32 --k 11 --len 64000 --batch 8 --act 1 --res 1 2>&1 | grep cin
