#!/usr/bin/env python3
"""The channel product of a frequency-domain conv alone (Conv1d._fft_product: ONE batched launch over the 64 bins) at the
Generator's stage shapes, hipGraph replay (HSP_FFT_PRODUCT=three | block selects hsp_cprod3_f32 or round 4's block matrix on
the conv kernel); for the block form, with --debug words, the kernel decomposition of the tuning build
(HSP_LIB=.../libhsp_tune.so: 1 producers stage chunk 0 only, 2 no MFMAs, 16 no epilogue, 2048 no barriers -- results are
then WRONG, the time says what the part costs).
    python tools/cprod_bench.py [--debug 0 1 2 16 17 2048] [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402
from megatts2_hierspeechpp_amd.hip_layers import Conv1d, finalize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--debug", type=int, nargs="*", default=[0])
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--k", type=int, default=11)
ap.add_argument("--d", type=int, default=1)
ap.add_argument("--stages", type=str, nargs="*", default=["512:800", "256:4000", "128:16000", "64:32000"])
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


print(f"product form: {hip_layers.FFT_PRODUCT}\nC L k d Np | debug word: ms (executed TF/s)")
for st in a.stages:
    C_, L_ = (int(v) for v in st.split(":"))
    lay = Conv1d(C_, C_, a.k, dilation=a.d, padding=(a.k - 1) * a.d // 2, weight_norm=True)
    lay.weight_v.data.normal_()
    lay.enable_fft()
    finalize(lay, dev)
    da = lay._fft_args(a.batch, L_)
    xf = torch.randn(64, 2 * C_, da.Np, device=dev)
    lay.ensure_wf()
    fl = 2.0 * 64 * (3 if lay._wf_form == "three" else 4) * C_ * C_ * da.Np   # executed: three C x C products or the 2C x 2C block
    line = f"{C_:4d} {L_:6d} {a.k:2d} {a.d} {da.Np:6d} |"
    for _ in range(200):                       # ~60 ms of the launch itself: clocks and caches as inside a step
        lay._fft_product(xf)
    torch.cuda.synchronize()
    for dbg in a.debug:
        hip_layers.DEBUG_FLAGS = dbg
        t = timed(lambda: lay._fft_product(xf))
        line += f" {dbg}: {t:.3f} ({fl / t / 1e9:5.1f})"
    hip_layers.DEBUG_FLAGS = 0
    print(line, flush=True)
    del lay, xf
    torch.cuda.empty_cache()
