#!/bin/bash
# whole-step A/B of scheduling knobs (ms per step of the default bench, hipGraph replay)
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" || exit 1; }
run HSP_FRONT_SPLITS=4
run HSP_FRONT_SPLITS=2
run HSP_FRONT_SPLITS=1
