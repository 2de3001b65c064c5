#!/usr/bin/env python3
"""Several conv shapes x tuning words in ONE process (kernel decomposition; needs the tuning build:
   HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so python tools/conv_sweep.py [--shapes ...] [--debug 0 1 2 16 17]
A shape is C:L:K[:dil[:res[:B[:Cout]]]].  Prints us per launch and algorithmic TFLOP/s (debug != 0: results are wrong,
the time is what is being measured)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="+", default=["256:4000:3", "256:4000:11", "128:16000:3", "128:16000:7", "512:800:3",
                                                "512:800:11", "64:32000:3", "64:32000:11", "32:64000:3", "32:64000:11"])
ap.add_argument("--debug", nargs="+", type=int, default=[0])
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rotate", type=int, default=1,
                help="(round 6) cycle the launches over this many separate (x, res, out) buffer sets, so that a launch finds "
                     "neither its input nor its residual in the Infinity Cache / L2 -- what a conv meets inside the step")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for shp in a.shapes:
    f = [int(v) for v in shp.split(":")]
    C_, L, K = f[:3]
    dil = f[3] if len(f) > 3 else 1
    res_on = f[4] if len(f) > 4 else 1
    B = f[5] if len(f) > 5 else 32
    Co = f[6] if len(f) > 6 else C_
    conv = hip_layers.Conv1d(C_, Co, K, dilation=dil, padding=(K - 1) * dil // 2)
    conv.weight.data.normal_(0, 0.05)
    conv.bias.data.normal_(0, 0.1)
    hip_layers.finalize(conv, dev)
    sets = [(torch.randn(B, C_, L, device=dev), torch.randn(B, Co, L, device=dev) if res_on else None,
             torch.empty(B, Co, L, device=dev)) for _ in range(a.rotate)]
    x, res, out = sets[0]
    fl = 2.0 * B * C_ * Co * K * L
    row = []
    for dbg in a.debug:
        hip_layers.DEBUG_FLAGS = dbg
        try:
            for _ in range(3):
                conv(x, res=res, out=out)
        except Exception:
            row.append(f"dbg{dbg}: n/a")
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.reps):
            x, res, out = sets[i % a.rotate]
            conv(x, res=res, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        row.append(f"dbg{dbg}: {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF")
    print(f"C {C_:4d}>{Co:4d} L {L:6d} K {K:2d} d {dil} res {res_on} B {B:2d} | " + " | ".join(row), flush=True)
