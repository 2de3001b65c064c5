set -x
R=$GRAFT_REPO_ROOT
cd $R
export HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so
for fs in 2 3 4; do
HSP_FRONT_SPLITS=$fs python tools/step_ab.py --debug 0 134217728 268435456 --rounds 2 --json gpurun_out/r05_ab_s64g2_fs$fs.json 2>&1 | grep round
done
