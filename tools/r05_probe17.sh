cd $GRAFT_REPO_ROOT
HSP_LIB=$GRAFT_REPO_ROOT/megatts2_hierspeechpp_amd/libhsp_tune.so python tools/b1_ab.py --debug 0 134217728 2>&1 | grep -v amdgpu
