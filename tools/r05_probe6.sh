set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python tools/plm_overlap_probe.py --cus 16 32 64 --json gpurun_out/r05_plm_overlap_probe.json > gpurun_out/r05_plm_overlap_probe.txt 2>&1
cat gpurun_out/r05_plm_overlap_probe.txt | grep -v amdgpu.ids
