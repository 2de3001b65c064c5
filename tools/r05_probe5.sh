set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r05_t_all.log 2>&1
tail -5 gpurun_out/r05_t_all.log
grep -q "passed" gpurun_out/r05_t_all.log && ! grep -q "failed\|error" gpurun_out/r05_t_all.log || exit 1
timeout -k 10 900 python tools/fftconv_table.py > gpurun_out/r05_fftconv_dispatch_table.txt 2> gpurun_out/r05_fftconv_dispatch_table.err
tail -25 gpurun_out/r05_fftconv_dispatch_table.txt
