#!/usr/bin/env python3
"""Same-box A/B of two BUILDS of the library on the headline step (or any command that prints bench.py's line):

    python tools/lib_ab.py --libs /path/libhsp_a.so /path/libhsp_b.so --rounds 3 [--json out.json] [-- extra bench.py args]

Each round runs `python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 20` once per library (HSP_LIB
selects it) as a child process, alternating A, B, A, B ... -- boxes and clocks drift, only alternating same-box
numbers are comparable.  This process never touches the GPU."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--libs", nargs="+", required=True)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--json", default=None)
ap.add_argument("--roofline", action="store_true", help="keep bench.py's per-launch pass (roofline.frac per library)")
ap.add_argument("rest", nargs="*")
a = ap.parse_args()
out = {lib: [] for lib in a.libs}
frac = {lib: [] for lib in a.libs}
for r in range(a.rounds):
    for lib in a.libs:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-cpu-baseline", "--steps", str(a.steps)]
        if not a.roofline:
            cmd.append("--no-roofline")
        cmd += a.rest
        p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, HSP_LIB=os.path.abspath(lib)))
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
        if p.returncode != 0 or not lines:
            sys.stderr.write(p.stderr[-2000:])
            raise SystemExit(f"bench.py failed with {lib}")
        d = json.loads(lines[-1])
        out[lib].append(d["ms_per_step"])
        if "roofline" in d:
            frac[lib].append(d["roofline"]["frac"])
        print(f"round {r} {os.path.basename(lib):24s} {d['ms_per_step']:.2f} ms / step"
              + (f"  frac {d['roofline']['frac']:.3f}" if "roofline" in d else ""), flush=True)
for lib in a.libs:
    v = sorted(out[lib])
    print(f"{os.path.basename(lib):24s} median {v[len(v) // 2]:.2f} ms  min {v[0]:.2f}  all {['%.2f' % x for x in out[lib]]}")
if a.json:
    with open(a.json, "w") as fh:
        json.dump({"ms_per_step": {os.path.basename(k): v for k, v in out.items()},
                   "roofline_frac": {os.path.basename(k): v for k, v in frac.items()}, "steps": a.steps,
                   "workload": "bench.py VocoderWorkload 32 x 4 s, hipGraph replay, one child process per run"}, fh, indent=1)
