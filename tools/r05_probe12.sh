set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "three_product or golden_frequency or layout_only" > gpurun_out/r05_t_cprod3.log 2>&1
tail -3 gpurun_out/r05_t_cprod3.log
grep -q "passed" gpurun_out/r05_t_cprod3.log && ! grep -q "failed" gpurun_out/r05_t_cprod3.log || exit 1
python tools/lib_ab.py --libs megatts2_hierspeechpp_amd/libhsp_prev.so megatts2_hierspeechpp_amd/libhsp.so --rounds 3 --json gpurun_out/r05_ab_cprod3_xcd.json
