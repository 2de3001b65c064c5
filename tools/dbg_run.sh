R=$GRAFT_REPO_ROOT
timeout 600 python3 $R/tools/gpu_check.py 2>&1 | grep -v "PASS" | tail -5
timeout 600 python3 -m pytest $R/tests -m gpu -x -q -k "transpose or golden" 2>&1 | tail -2
timeout 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-launches $R/gpurun_out/launches9.txt 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])"
grep " 3 |" $R/gpurun_out/launches9.txt
