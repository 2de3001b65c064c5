set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r05_t_all.log 2>&1
tail -3 gpurun_out/r05_t_all.log
grep -q "passed" gpurun_out/r05_t_all.log && ! grep -q "failed\|error" gpurun_out/r05_t_all.log || exit 1
export HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so
python tools/b1_ab.py --debug 0 134217728 2>&1 | grep -v amdgpu > gpurun_out/r05_b1_ab.txt; cat gpurun_out/r05_b1_ab.txt
python tools/step_ab.py --debug 0 134217728 --rounds 3 --json gpurun_out/r05_ab_ksplit_step.json 2>&1 | grep round
