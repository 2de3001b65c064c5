#!/usr/bin/env python3
"""Direct MFMA conv against its frequency-domain form (Conv1d.forward_fft: forward DFT, batched 1x1 over 64 bins, inverse
DFT) on the AMP-block shapes of the Generator, B = 32, hipGraph replay; with --stages the three launches separately.
    python tools/fftconv_bench.py [--stages]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402
from megatts2_hierspeechpp_amd.hip_layers import Conv1d, finalize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stages", action="store_true")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--k", type=int, nargs="*", default=[11, 7, 3])
ap.add_argument("--d", type=int, nargs="*", default=[1, 3, 5])
a = ap.parse_args()
dev = torch.device("cuda:0")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps)


print("C L k d | direct ms (TF/s) | frequency-domain ms (algorithmic TF/s) | ratio" + (" | fwd / product / inv ms" if a.stages else ""))
for C_, L_ in ((512, 800), (256, 4000), (128, 16000), (64, 32000)):
    for k in a.k:
        for d in a.d:
            lay = Conv1d(C_, C_, k, dilation=d, padding=(k - 1) * d // 2, weight_norm=True)
            lay.enable_fft()
            finalize(lay, dev)
            x = torch.randn(a.batch, C_, L_, device=dev)
            res = torch.randn_like(x)
            out = torch.empty_like(x)
            t_d = timed(lambda: lay(x, res=res, out=out))
            t_f = timed(lambda: lay.forward_fft(x, res=res, out=out))
            fl = 2.0 * a.batch * C_ * C_ * k * L_
            line = f"{C_:4d} {L_:6d} {k:2d} {d} | {t_d:7.3f} ({fl / t_d / 1e9:6.1f}) | {t_f:7.3f} ({fl / t_f / 1e9:6.1f}) | {t_d / t_f:5.2f}"
            if a.stages:
                rec = []
                hip_layers.LAUNCH_HOOK = lambda kind, fl_, nb, e0, e1, la: rec.append((kind, e0, e1))
                lay.forward_fft(x, res=res, out=out)
                torch.cuda.synchronize()
                hip_layers.LAUNCH_HOOK = None
                line += " | " + " / ".join(f"{e0.elapsed_time(e1):.3f}" for _, e0, e1 in rec)
            print(line, flush=True)
            del lay, x, res, out
            torch.cuda.empty_cache()
