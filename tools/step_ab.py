#!/usr/bin/env python3
"""Same-box A/B of the whole vocoder step (bench.py's workload, hipGraph replay) under several values of the conv
kernel's tuning word -- needs the tuning build:
    HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so python tools/step_ab.py --debug 0 32768 --rounds 2 [--json out.json]
Variants alternate (boxes and clocks drift); prints ms per step per round."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--debug", nargs="+", type=int, default=[0])
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--json", default=None)
a = ap.parse_args()
args = bench.parse_args(["--steps", str(a.steps)])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.VocoderWorkload(args, 0, 1, dev)
wl.model.finalize(dev)
wl.prepare(0, args.batch)
steps = {}
for d in a.debug:
    hip_layers.DEBUG_FLAGS = d
    steps[d] = wl.make_step()
out = {d: [] for d in a.debug}
for r in range(a.rounds):
    for d in a.debug:
        st = steps[d]
        for _ in range(3):
            st()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            st()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        out[d].append(ms)
        print(f"round {r} debug {d:6d}: {ms:.2f} ms / step", flush=True)
if a.json:
    with open(a.json, "w") as fh:
        json.dump({"ms_per_step_by_debug_word": {str(k): v for k, v in out.items()}, "steps": a.steps,
                   "workload": "bench.py VocoderWorkload 32 x 4 s, hipGraph replay"}, fh, indent=1)
