#!/usr/bin/env python3
"""In-kernel cycle-counter stamps of workgroup 0 of bgemm_kernel (tuning build only):
    HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so python tools/bgemm_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402

dev = torch.device("cuda:0")
orig = hip_layers._launch
st = torch.zeros(8, dtype=torch.int64, device=dev)


def launch(kind, fn, a, flops, nbytes, soft=False, keep=()):
    a.filt = st.data_ptr()
    return orig(kind, fn, a, flops, nbytes, soft, keep)


hip_layers._launch = launch
for tile, bit in (("128x128", 1 << 18), ("128x64", 1 << 19), ("64x64", 1 << 23)):
    for extra in (0, 1, 2):
        hip_layers.DEBUG_FLAGS = (1 << 20) | bit | extra
        for K, N, M in ((276, 128, 128), (276, 3200, 1104), (1104, 3200, 276)):
            lin = hip_layers.LinearCT(K, M)
            lin.weight.data.normal_(0, 0.05)
            hip_layers.finalize(lin, dev)
            x = torch.randn(1, K, N, device=dev)
            for _ in range(3):
                lin(x)
            torch.cuda.synchronize()
            s = st.cpu().tolist()
            print(f"{tile} dbg+{extra} K {K:4d} N {N:4d} M {M:4d}: total {s[3] - s[0]:6d} | start -> stage 0 in LDS {s[1] - s[0]} | main loop "
                  f"{s[2] - s[1]} | epilogue {s[3] - s[2]} || producer: start -> first issue {s[4] - s[0]} | issue of the prologue "
                  f"{s[5] - s[4]} | wait for stage 0 {s[6] - s[5]}", flush=True)
