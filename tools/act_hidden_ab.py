#!/usr/bin/env python3
"""How much of the stand-alone activation's kernel time does the STREAMED step actually pay?  
The 128 act1d_seg_kernel launches of a step take ~7.3 ms when issued one after the other (bench.py's serial per-launch
pass: 0.60-0.62 of the HBM peak), but in the timed step the three AMP chains of a stage run on three streams, so a
chain's activation overlaps the other chains' convs.  Same-box A/B: the step's hipGraph with every activation launch
removed (this script swaps functional.act1d for the identity while it captures: wrong audio, same conv work; only the
stand-alone launches -- the activations fused into forward transforms and pair launches stay) against the real one, alternating.
    python tools/act_hidden_ab.py [--rounds 3] [--json out.json]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from megatts2_hierspeechpp_amd import functional as Fh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
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
_act1d = Fh.act1d          # the knob lives HERE (round 5): the product module has no switch that turns an op into the identity
for name, skip in (("with activations", False), ("activations removed", True)):
    Fh.act1d = (lambda x, ea, binv, filt, out=None: x.contiguous()) if skip else _act1d
    steps[name] = wl.make_step()
Fh.act1d = _act1d
out = {k: [] for k in steps}
for r in range(a.rounds):
    for name, st in steps.items():
        for _ in range(3):
            st()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            st()
        torch.cuda.synchronize()
        out[name].append(1e3 * (time.perf_counter() - t0) / a.steps)
        print(f"round {r} {name:20s}: {out[name][-1]:.2f} ms / step", flush=True)
med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
paid = med["with activations"] - med["activations removed"]
print(f"the streamed step pays {paid:.2f} ms for its 128 activation launches")
if a.json:
    with open(a.json, "w") as fh:
        json.dump({"ms_per_step": out, "median": med, "activation_ms_paid_by_the_streamed_step": paid,
                   "workload": "bench.py VocoderWorkload 32 x 4 s, hipGraph replay; 'removed' = identity instead of the "
                               "activation launch (wrong audio, tools only)"}, fh, indent=1)
