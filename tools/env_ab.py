#!/usr/bin/env python3
"""Same-box A/B of the headline step under several ENVIRONMENT settings (host-side policy knobs such as HSP_GEN_GROUPS,
HSP_FRONT_SPLITS, HSP_FFT_PAIR), alternating child processes as tools/lib_ab.py does for two builds of the library:

    python tools/env_ab.py --env "HSP_GEN_GROUPS=1" "HSP_GEN_GROUPS=2" "HSP_GEN_GROUPS=4" --rounds 2 [--roofline] [--json out.json]

An entry may set several variables ("A=1,B=2") or none ("" = the defaults).  With --roofline the per-launch pass of
bench.py stays on and the stand-alone activation's GB/s (roofline_activation) is reported beside the step time.  This
process never touches the GPU."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--env", nargs="+", required=True)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--roofline", action="store_true")
ap.add_argument("--json", default=None)
ap.add_argument("rest", nargs="*")
a = ap.parse_args()
res = {e: {"ms_per_step": [], "act_gbs": [], "act_ms": [], "frac": []} for e in a.env}
for r in range(a.rounds):
    for e in a.env:
        env = dict(os.environ)
        for kv in filter(None, e.split(",")):
            k, v = kv.split("=", 1)
            env[k] = v
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-cpu-baseline", "--steps", str(a.steps)]
        if not a.roofline:
            cmd.append("--no-roofline")
        cmd += a.rest
        p = subprocess.run(cmd, capture_output=True, text=True, env=env)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
        if p.returncode != 0 or not lines:
            sys.stderr.write(p.stderr[-3000:])
            raise SystemExit(f"bench.py failed under {e!r}")
        d = json.loads(lines[-1])
        res[e]["ms_per_step"].append(d["ms_per_step"])
        extra = ""
        if "roofline" in d:
            res[e]["frac"].append(d["roofline"]["frac"])
            ra = d.get("roofline_activation") or {}
            if ra:
                res[e]["act_gbs"].append(ra.get("achieved"))
                res[e]["act_ms"].append(ra.get("kernel_ms_per_step"))
                extra = f"  conv frac {d['roofline']['frac']:.3f}  activation {ra.get('achieved', 0):.0f} GB/s, {ra.get('kernel_ms_per_step', 0):.2f} ms"
        print(f"round {r} {e or '(defaults)':40s} {d['ms_per_step']:.2f} ms / step{extra}", flush=True)
for e in a.env:
    v = sorted(res[e]["ms_per_step"])
    print(f"{e or '(defaults)':40s} median {v[len(v) // 2]:.2f} ms  min {v[0]:.2f}  all {['%.2f' % x for x in res[e]['ms_per_step']]}")
if a.json:
    with open(a.json, "w") as fh:
        json.dump({"results": res, "steps": a.steps,
                   "workload": "bench.py VocoderWorkload 32 x 4 s, hipGraph replay, one child process per run"}, fh, indent=1)
