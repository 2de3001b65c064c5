#!/usr/bin/env python3
"""In-kernel phase stamps (s_memtime) of workgroup 0 of mha_tok_kernel (tuning build only):
    HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so python tools/mha_stamps.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
B, H, D = 16, 4, 69
for T in (16, 64, 128, 200):
    qkv = torch.randn(3 * H * D, B * T, device=dev)
    o = torch.empty(H * D, B * T, device=dev)
    per = lambda m: m.reshape(-1, B, T).permute(1, 0, 2)
    q, k, v = (per(qkv[i * H * D:(i + 1) * H * D]) for i in range(3))
    oo = per(o)
    st = torch.zeros(8, dtype=torch.int64, device=dev)
    a = L.MhaArgs()
    a.q, a.k, a.v, a.o = L.fptr(q), L.fptr(k), L.fptr(v), L.fptr(oo)
    a.q_bs, a.k_bs, a.v_bs, a.o_bs = q.stride(0), k.stride(0), v.stride(0), oo.stride(0)
    a.q_cs, a.k_cs, a.v_cs, a.o_cs = q.stride(1), k.stride(1), v.stride(1), oo.stride(1)
    a.B, a.H, a.D, a.Tq, a.Tk = B, H, D, T, T
    a.qk_scale = D ** -0.5
    a.rel_v, a.window = st.data_ptr(), 1003
    for _ in range(3):
        L.check(L.lib().hsp_mha_f32(C.byref(a), L.stream_ptr()), "hsp_mha_f32")
    torch.cuda.synchronize()
    s = st.cpu().tolist()
    d = [(s[i + 1] - s[i]) for i in range(7)]
    names = ["scores (loads + MFMA + S store)", "barrier", "V request + softmax", "barrier", "PV", "barrier", "store"]
    print(f"T = {T:3d}: total {s[7] - s[0]} ticks (100 MHz: {(s[7] - s[0]) / 100:.1f} us) | " + " | ".join(f"{n} {x}" for n, x in zip(names, d)))
