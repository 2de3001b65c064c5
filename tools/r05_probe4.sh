set -x
R=$GRAFT_REPO_ROOT
cd $R
export HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so
python tools/cprod_bench.py --debug 0 4096 0 1 16 17 2048 0 > gpurun_out/r05_cprod3_decomp.txt 2>&1
python tools/cprod_bench.py --d 5 --debug 0 4096 0 --stages 512:800 >> gpurun_out/r05_cprod3_decomp.txt 2>&1
python tools/cprod_bench.py --d 3 --debug 0 4096 0 --stages 512:800 >> gpurun_out/r05_cprod3_decomp.txt 2>&1
cat gpurun_out/r05_cprod3_decomp.txt
python tools/step_ab.py --debug 0 67108864 --rounds 3 --json gpurun_out/r05_ab_halfcu.json
python tools/stage_split.py > gpurun_out/r05_stage_split.txt 2>&1; tail -20 gpurun_out/r05_stage_split.txt
