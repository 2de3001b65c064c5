set -x
R=$GRAFT_REPO_ROOT
cd $R
export HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so
python tools/cprod_bench.py --debug 0 17 8209 0 17 8209 8192 --stages 256:4000 128:16000 > gpurun_out/r05_cprod3_valu.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_cprod3_valu.txt
