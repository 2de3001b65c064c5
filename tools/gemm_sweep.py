#!/usr/bin/env python3
"""Is the 1x1 / nn.Linear dispatcher (hsp_conv1d_mfma_f32 -> bgemm / rgemm / tokgemm / conv tiles) within a few per cent
of the best kernel at EVERY (B, T), not only at the two workloads its thresholds were fitted to?  

    HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so python tools/gemm_sweep.py > profiles/r04_gemm_dispatch_table.txt

For the 1x1 shapes of the product path -- the PLM layer's four GEMMs (ttv_v1/transformer_mega.py:54-61,108-113; columns
= B * T tokens of one [C, B*T] matrix) and the DiT / WN 1x1s of the vocoder's 50 Hz part (modules.py:166-174,357-411;
[B, C, T] tensors) -- at B in {1, 4, 8, 16, 32, 64} x T in {50, 200, 1000}: the dispatcher's own choice against every
kernel FORCED through the tuning word (hipGraph replay of back-to-back launches; a forced kernel that does not take
the shape reads n/a)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L  # noqa: E402
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402
from megatts2_hierspeechpp_amd.ttv_v1.transformer_mega import LayerNorm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--quick", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
NO_BG = 1 << 22
VARIANTS = [("dispatcher", 0), ("bgemm 64x64", 1 << 23), ("bgemm 128x128", 1 << 18), ("rgemm 64x32", NO_BG | (1 << 24) | 8),
            ("rgemm 32x32", NO_BG | (1 << 24) | 16), ("tokgemm", NO_BG | (1 << 24) | 131072), ("conv tiles", 128)]
NAMES = {-2: "bgemm", -1: "rgemm", 0: "tokgemm"}
# (label, Cin, Cout, fused LayerNorm, residual, activation, layout)
SHAPES = [("PLM qkv (LN)", 276, 828, 1, 0, L.ACT_NONE, "cols"), ("PLM out_proj", 276, 276, 0, 1, L.ACT_NONE, "cols"),
          ("PLM ff.0 (LN, ReLU)", 276, 1104, 1, 0, L.ACT_RELU, "cols"), ("PLM ff.3", 1104, 276, 0, 1, L.ACT_NONE, "cols"),
          ("DiT qkv", 192, 576, 0, 0, L.ACT_NONE, "batch"), ("DiT proj", 192, 192, 0, 1, L.ACT_NONE, "batch"),
          ("DiT fc2", 768, 192, 0, 1, L.ACT_NONE, "batch"), ("WN res_skip", 192, 384, 0, 0, L.ACT_NONE, "batch")]
BS, TS = ([8, 32], [200]) if a.quick else ([1, 4, 8, 16, 32, 64], [50, 200, 1000])


def timed(fn):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(a.reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * a.reps) * 1e3


print("shape | B x T | dispatcher: kernel us | " + " | ".join(n for n, _ in VARIANTS[1:]) + " | best | dispatcher / best")
worst = 0.0
for label, K, M, ln, res_on, act, layout in SHAPES:
    class Mod(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.norm = LayerNorm(K)
            self.lin = hip_layers.LinearCT(K, M)
            if ln:
                self.lin.fuse_input_layernorm(self.norm)

    m = Mod()
    m.lin.weight.data.normal_(0, 0.05)
    hip_layers.finalize(m, dev)
    for B in BS:
        for T in TS:
            shp = (1, K, B * T) if layout == "cols" else (B, K, T)
            x = torch.randn(*shp, device=dev)
            out = torch.empty(shp[0], M, shp[2], device=dev)
            res = torch.randn_like(out) if res_on else None
            plans = []

            def hook(kind, fl, nb, e0, e1, la):
                plan = (C.c_int32 * 4)()
                L.check(L.lib().hsp_conv1d_mfma_plan(C.byref(la), C.byref(plan)), "plan")
                plans.append(tuple(plan))

            times = []
            for name, dbg in VARIANTS:
                hip_layers.DEBUG_FLAGS = dbg
                try:
                    if dbg == 0:
                        hip_layers.LAUNCH_HOOK = hook
                        m.lin(x, res=res, out=out, act=act)
                        hip_layers.LAUNCH_HOOK = None
                    times.append(timed(lambda: m.lin(x, res=res, out=out, act=act)))
                except Exception:  # noqa: BLE001  (the forced kernel does not take this shape)
                    hip_layers.LAUNCH_HOOK = None
                    times.append(None)
            hip_layers.DEBUG_FLAGS = 0
            p = plans[-1] if plans else (0, 0, 1, 0)
            kern = NAMES.get(p[2], "conv tiles") + f" {p[0]}x{p[1]}"
            ok = [(t, n) for t, (n, _) in zip(times[1:], VARIANTS[1:]) if t is not None]
            best_t, best_n = min(ok) if ok else (times[0], "dispatcher")
            ratio = times[0] / best_t
            worst = max(worst, ratio)
            print(f"{label:20s} | {B:2d} x {T:4d} | {kern:18s} {times[0]:7.1f} | " +
                  " | ".join("   n/a" if t is None else f"{t:6.1f}" for t in times[1:]) +
                  f" | {best_n:13s} | {ratio:5.2f}" + ("  <-- > 5 %" if ratio > 1.05 else ""), flush=True)
print(f"worst dispatcher / best: {worst:.2f}")
