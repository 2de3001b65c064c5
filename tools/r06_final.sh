# round 6 final measurements (one gpurun call): the whole GPU test suite + smoke(), rocprof kernel stats of the serialised
# eager step + its bench line, PMC traffic passes (vocoder step; TTS / SR48 / wav2vec2 / denoiser stages), the transform
# kernels' counters, the driver-format bench line, the per-shape launch table
set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r06_t_all.log 2>&1
tail -3 gpurun_out/r06_t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -2 gpurun_out/r06_smoke.log
bash tools/profile_final.sh && cd $R &&
bash tools/pmc_traffic.sh && cd $R &&
bash tools/pmc_traffic_extra.sh && cd $R &&
bash tools/pmc_dftseg.sh; cd $R
python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
tail -c 300 gpurun_out/r06_bench_final.json
python bench.py --dump-launches gpurun_out/r06_launch_table_final.txt --no-extra --no-cpu-baseline > gpurun_out/r06_bench_second.json 2>/dev/null
head -30 gpurun_out/r06_launch_table_final.txt
