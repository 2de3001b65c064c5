R=$GRAFT_REPO_ROOT
timeout 600 python3 $R/tools/gpu_check.py 2>&1 | grep -v PASS | tail -5
for shape in "--cin 32 --cout 32 --k 7 --len 64000 --batch 8" "--cin 128 --cout 128 --k 7 --len 16000 --batch 8" "--cin 512 --cout 512 --k 11 --len 1024 --batch 16"; do
for dbg in 0 3; do
  python3 $R/tools/conv_bench.py $shape --debug $dbg --res 1 2>&1 | grep cin
done; done
timeout 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-launches $R/gpurun_out/launches6.txt 2>&1 | tail -1 | cut -c1-330
