set -x
R=$GRAFT_REPO_ROOT
cd $R
export HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so
python tools/conv_sweep.py --shapes 192:200:5:1:0:8:768 192:200:5:1:0:32:768 192:200:5:1:0:8:384 --debug 0 1 2 3 16 17 2048 2049 0 --reps 50 > gpurun_out/r05_front_decomp.txt 2>&1
python tools/wn_bench.py --batch 8 --debug 0 1 2 3 >> gpurun_out/r05_front_decomp.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_front_decomp.txt
