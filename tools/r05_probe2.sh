# round 5, second GPU call: the three-product channel mix -- parity tests, per-launch bench against the block form, step A/B
set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -s -k "weight_spectrum or dynamic_range" > gpurun_out/r05_t_cprod3.log 2>&1
tail -15 gpurun_out/r05_t_cprod3.log
grep -q "passed" gpurun_out/r05_t_cprod3.log && ! grep -q "failed" gpurun_out/r05_t_cprod3.log || exit 1
for d in 1 5; do
HSP_FFT_PRODUCT=three python tools/cprod_bench.py --d $d > gpurun_out/r05_cprod_three_d$d.txt 2>&1
HSP_FFT_PRODUCT=block python tools/cprod_bench.py --d $d > gpurun_out/r05_cprod_block_d$d.txt 2>&1
cat gpurun_out/r05_cprod_three_d$d.txt gpurun_out/r05_cprod_block_d$d.txt
done
for r in 1 2; do
HSP_FFT_PRODUCT=block python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/r05_ab_block_$r.json 2>/dev/null
HSP_FFT_PRODUCT=three python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/r05_ab_three_$r.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_ab_*.json')):
    try: print(f, json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
done
