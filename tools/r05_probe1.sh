# round 5, first GPU call: baseline bench of the round-4 sources on this box, decomposition of the K = 1 channel product
# (tuning build), counters of the transform kernels
set -x
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --no-extra > gpurun_out/r05_bench_base.json 2> gpurun_out/r05_bench_base.err
tail -c 600 gpurun_out/r05_bench_base.json
HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so python tools/cprod_bench.py --debug 0 1 2 3 16 17 2048 2049 > gpurun_out/r05_cprod_decomp.txt 2>&1
cat gpurun_out/r05_cprod_decomp.txt
bash tools/pmc_dftseg.sh
