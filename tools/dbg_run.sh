R=$GRAFT_REPO_ROOT
timeout 600 python3 $R/tools/gpu_check.py 2>&1 | grep -v PASS | tail -5
timeout 600 python3 -m pytest $R/tests -m gpu -x -q 2>&1 | tail -2
for dbg in 0 64; do
python3 $R/tools/conv_bench.py --cin 256 --cout 256 --k 3 --len 4096 --batch 8 --act 0 --debug $dbg 2>&1 | grep cin
python3 $R/tools/conv_bench.py --cin 128 --cout 128 --k 7 --len 16000 --batch 8 --act 0 --debug $dbg 2>&1 | grep cin
done
timeout 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
