#!/usr/bin/env python3
"""hsp_mha_proj_f32 (attention over all heads + projection, one launch) against the two launches it replaces
(hsp_mha_f32 + the projection GEMM), PLM layout, hipGraph replay of a chain of launches with an L2-dirtying kernel in
between (as in the loop: nothing of a launch's operands is in L2 when it starts); with the tuning build also the
in-kernel phase stamps of the middle workgroup:
    [HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so] python tools/mhaproj_bench.py [--dit]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L  # noqa: E402
from megatts2_hierspeechpp_amd import functional as Fh  # noqa: E402
from megatts2_hierspeechpp_amd.hip_layers import LinearCT, finalize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dit", action="store_true")
ap.add_argument("--batch", type=int, default=None)
a_ = ap.parse_args()
dev = torch.device("cuda:0")
H, D = (2, 96) if a_.dit else (4, 69)
B = a_.batch or (8 if a_.dit else 16)
Cc = H * D
lin = LinearCT(Cc, Cc)
lin.keep_rowmajor_weight()
lin.weight.data = torch.randn(Cc, Cc) / Cc ** 0.5
finalize(lin, dev)
tuning = "tune" in os.path.basename(L.LIB_PATH)
scratch = torch.zeros(16 << 20, device=dev)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            scratch.mul_(1.0001)       # 64 MB through L2 between launches
            fn()
    gd = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gd):
        for _ in range(n):
            scratch.mul_(1.0001)
    res = []
    for gg in (g, gd):
        gg.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gg.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 / n * 1e3)
    return res[0] - res[1]


print(f"H={H} D={D} B={B} lib={os.path.basename(L.LIB_PATH)}")
for T in (8, 16, 40, 64, 100, 128, 160, 200):
    Np = (B * T + 3) & ~3
    qkv = torch.randn(1, 3 * Cc, Np, device=dev)
    x = torch.randn(1, Cc, Np, device=dev)
    per = lambda m: m[:, :B * T].reshape(-1, B, T).permute(1, 0, 2)
    q, k, v = (per(qkv[0, i * Cc:(i + 1) * Cc]) for i in range(3))
    y = torch.empty_like(x)
    o = torch.empty_like(x)
    scale = D ** -0.5

    def fused():
        Fh.mha_proj(q, k, v, H, scale, lin._wt, bias=lin._b, res=per(x[0]), out=per(y[0]))

    def two():
        Fh.mha(q, k, v, H, scale, out=per(o[0]))
        lin(o, res=x, out=y)

    t_f, t_2 = timed(fused), timed(two)
    line = f"T = {T:3d}: fused {t_f:6.1f} us   attention + projection {t_2:6.1f} us"
    if tuning:
        st = torch.zeros(10, dtype=torch.int64, device=dev)
        a = L.MhaProjArgs()
        a.q, a.k, a.v = L.fptr(q), L.fptr(k), L.fptr(v)
        a.q_bs, a.q_cs, a.k_bs, a.k_cs, a.v_bs, a.v_cs = q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1)
        a.B, a.H, a.D, a.Tq, a.Tk, a.qk_scale = B, H, D, T, T, scale
        a.wt, a.M, a.wt_ld, a.bias = L.fptr(lin._wt), Cc, Cc, L.fptr(lin._b)
        r, yy = per(x[0]), per(y[0])
        a.res, a.res_bs, a.res_cs, a.res_ts = L.fptr(r), r.stride(0), r.stride(1), 1
        a.y, a.y_bs, a.y_cs, a.y_ts = L.fptr(yy), yy.stride(0), yy.stride(1), 1
        a.cscale, a.debug = st.data_ptr(), 1
        for _ in range(3):
            scratch.mul_(1.0001)
            L.check(L.lib().hsp_mha_proj_f32(C.byref(a), L.stream_ptr()), "hsp_mha_proj_f32")
        torch.cuda.synchronize()
        s = st.cpu().tolist()
        names = ["W req + Q", "QK", "softmax", "PV", "barrier", "merge", "proj", "epilogue"]
        line += "  | stamps (100 MHz ticks): " + " ".join(f"{n} {s[i + 1] - s[i]}" for i, n in enumerate(names)) + f" total {s[8] - s[0]}"
    print(line, flush=True)
