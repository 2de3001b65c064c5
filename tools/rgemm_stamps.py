#!/usr/bin/env python3
"""In-kernel cycle-counter stamps of one workgroup of rgemm_kernel (tuning build only):
    HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so python tools/rgemm_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L, hip_layers  # noqa: E402

dev = torch.device("cuda:0")
orig = hip_layers._launch
st = torch.zeros(8, dtype=torch.int64, device=dev)


def launch(kind, fn, a, flops, nbytes, soft=False, keep=()):
    a.filt = st.data_ptr()
    return orig(kind, fn, a, flops, nbytes, soft, keep)


hip_layers._launch = launch
hip_layers.DEBUG_FLAGS = 1 << 20
for K, N, M in ((276, 160, 828), (276, 1600, 828), (276, 1600, 276), (1104, 1600, 276), (1104, 160, 276), (276, 16, 1024)):
    lin = hip_layers.LinearCT(K, M)
    lin.weight.data.normal_(0, 0.05)
    hip_layers.finalize(lin, dev)
    x = torch.randn(1, K, N, device=dev)
    for _ in range(3):
        lin(x)
    torch.cuda.synchronize()
    s = st.cpu().tolist()
    d = [s[i + 1] - s[i] for i in range(4)]
    print(f"K {K:4d} N {N:4d} M {M:4d}: total {s[4] - s[0]:6d} cycles | operand loads + MFMA {d[0]} | LDS write {d[1]} | barrier {d[2]} | reduce + epilogue store {d[3]}")
