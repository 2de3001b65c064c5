set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "three_product or golden_frequency or layout_only or falls_back" > gpurun_out/r05_t_cprod3.log 2>&1
tail -5 gpurun_out/r05_t_cprod3.log
grep -q "passed" gpurun_out/r05_t_cprod3.log && ! grep -q "failed" gpurun_out/r05_t_cprod3.log || exit 1
HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so python tools/cprod_bench.py --debug 0 1 2 3 16 17 2048 2049 4096 > gpurun_out/r05_cprod3_decomp.txt 2>&1
HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so python tools/cprod_bench.py --d 5 --debug 0 4096 --stages 512:800 >> gpurun_out/r05_cprod3_decomp.txt 2>&1
HSP_LIB=$R/megatts2_hierspeechpp_amd/libhsp_tune.so python tools/cprod_bench.py --d 3 --debug 0 4096 --stages 512:800 >> gpurun_out/r05_cprod3_decomp.txt 2>&1
cat gpurun_out/r05_cprod3_decomp.txt
python bench.py --no-extra --no-cpu-baseline --steps 20 > gpurun_out/r05_bench_cprod3.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_cprod3.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['channel_products'], d['roofline'].get('mfma_gemm_population'), d['derive_ms'], d['derived_mb'], d['pack_ms'], d['config']['weights_mb'])
print(d['roofline']['by_tile_shape'])
PY
