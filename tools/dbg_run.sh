R=$GRAFT_REPO_ROOT
timeout 600 python3 -m pytest $R/tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
