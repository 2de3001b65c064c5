set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or wn or WN or survey or drop_in or full_size or frequency or layout" > gpurun_out/r05_t_ks.log 2>&1
tail -3 gpurun_out/r05_t_ks.log
grep -q "passed" gpurun_out/r05_t_ks.log && ! grep -q "failed\|error" gpurun_out/r05_t_ks.log || exit 1
export HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so
python tools/wn_bench.py --batch 8 --debug 0 134217728 > gpurun_out/r05_wn_ks.txt 2>&1
python tools/wn_bench.py --batch 1 --debug 0 134217728 >> gpurun_out/r05_wn_ks.txt 2>&1
grep -v amdgpu gpurun_out/r05_wn_ks.txt
python tools/step_ab.py --debug 0 134217728 --rounds 3 --json gpurun_out/r05_ab_s64g2.json
