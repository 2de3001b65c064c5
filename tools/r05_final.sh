# round 5 final measurements: rocprof kernel stats of the serialised eager step + its bench line, PMC traffic passes, the
# dispatch table of the conv forms under the final policy, the driver-format bench line
set -x
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_final.sh && cd $R &&
bash tools/pmc_traffic.sh && cd $R &&
python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
tail -c 400 gpurun_out/r05_bench_final.json
timeout -k 10 900 python tools/fftconv_table.py > gpurun_out/r05_fftconv_dispatch_table.txt 2> gpurun_out/r05_fftconv_dispatch_table.err
tail -8 gpurun_out/r05_fftconv_dispatch_table.txt
