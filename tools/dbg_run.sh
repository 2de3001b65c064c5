R=$GRAFT_REPO_ROOT
timeout 600 python3 $R/tools/gpu_check.py 2>&1 | tail -26
for dbg in 0 1; do
for shape in "--cin 512 --cout 512 --k 11 --len 1024 --batch 16" "--cin 512 --cout 512 --k 3 --len 1024 --batch 16" "--cin 128 --cout 128 --k 11 --len 16000 --batch 8" "--cin 32 --cout 32 --k 7 --len 64000 --batch 8"; do
  python3 $R/tools/conv_bench.py $shape --debug $dbg 2>&1 | grep cin
done; done
