cd $GRAFT_REPO_ROOT
HSP_LIB=$GRAFT_REPO_ROOT/megatts2_hierspeechpp_amd/libhsp_tune.so timeout -k 5 120 python tools/ks_debug.py 2>&1 | grep -v amdgpu
