set -x
R=$GRAFT_REPO_ROOT
cd $R
python tools/plm_step_curve.py --batch 16 32 > gpurun_out/r05_plm_step_curve.txt 2>gpurun_out/r05_plm_step_curve.err
head -8 gpurun_out/r05_plm_step_curve.txt; grep "^#" gpurun_out/r05_plm_step_curve.txt
for b in 16 32 48; do python tools/plm_bench.py --batch $b 2>/dev/null | tail -2; done
python tools/tts_bench.py --batch 32 > gpurun_out/r05_tts_b32.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r05_tts_b32.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
timeout -k 10 600 python tools/plm_overlap_probe.py --cus 96 128 192 256 --json gpurun_out/r05_plm_overlap_probe2.json > gpurun_out/r05_plm_overlap_probe2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_plm_overlap_probe2.txt
python bench.py > gpurun_out/r05_bench_mid.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_mid.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:(v.get('ms_per_step'), v.get('stage_ms')) for k,v in d['extra_configs'].items()})
PY
