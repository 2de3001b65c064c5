#!/usr/bin/env python3
"""hsp_bgemm.hip against torch (fp64 accumulate) and against the other token-GEMM kernels, per forced tile shape.
    HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so python tools/bgemm_check.py
Tuning words: 1<<18 = 128 x 128, 1<<19 = 128 x 64, 1<<21 = 64 x 128, 1<<23 = 64 x 64, 1<<22 = the kernel switched off (old dispatch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L  # noqa: E402
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402
from megatts2_hierspeechpp_amd.ttv_v1.transformer_mega import LayerNorm  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
VARIANTS = {"128x128": 1 << 18, "128x64": 1 << 19, "64x128": 1 << 21, "64x64": 1 << 23, "old": 1 << 22, "default": 0}
worst = 0.0
# K, N, M, B, ln, res, act
CASES = [(276, 3200, 1104, 1, 1, 0, L.ACT_RELU), (276, 3200, 828, 1, 1, 0, 0), (276, 1600, 1104, 1, 1, 0, L.ACT_RELU),
         (276, 1604, 828, 1, 1, 0, 0), (276, 132, 276, 1, 0, 1, 0), (1104, 1000, 276, 1, 0, 1, 0), (192, 200, 576, 3, 0, 0, 0),
         (96, 64, 128, 2, 0, 1, L.ACT_GELU_TANH), (276, 4, 1104, 1, 1, 0, 0), (100, 260, 36, 1, 1, 1, 0)]
for K, N, M, B, ln, res_on, act in CASES:

    class Mod(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.norm = LayerNorm(K)
            self.lin = hip_layers.LinearCT(K, M)
            if ln:
                self.lin.fuse_input_layernorm(self.norm)

    m = Mod()
    m.lin.weight.data.normal_(0, 0.05)
    m.lin.bias.data.normal_(0, 0.5)
    m.norm.weight.data.uniform_(0.5, 1.5)
    m.norm.bias.data.normal_(0, 0.3)
    hip_layers.finalize(m, dev)
    x = (torch.randn(B, K, N, device=dev) * 1.7 + 3.0)          # |mean| > std: the case the pivot shift is for
    res = torch.randn(B, M, N, device=dev) if res_on else None
    xd = x.double()
    if ln:
        xd = torch.nn.functional.layer_norm(xd.transpose(1, 2), (K,), m.norm.weight.data.double().to(dev),
                                            m.norm.bias.data.double().to(dev), 1e-5).transpose(1, 2)
    ref = torch.einsum("mk,bkn->bmn", m.lin.weight.data.double().to(dev), xd) + m.lin.bias.data.double().to(dev)[None, :, None]
    if act == L.ACT_RELU:
        ref = ref.clamp_min(0)
    elif act == L.ACT_GELU_TANH:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if res is not None:
        ref = ref + res.double()
    row = []
    for name, dbg in VARIANTS.items():
        hip_layers.DEBUG_FLAGS = dbg
        out = torch.full((B, M, N), float("nan"), device=dev)
        try:
            m.lin(x, res=res, out=out, act=act)
            torch.cuda.synchronize()
            err = float((out.double() - ref).abs().max())
            row.append(f"{name} {err:.2e}")
            if name != "old":
                worst = max(worst, err if err == err else 1e9)
        except Exception as e:  # noqa: BLE001
            row.append(f"{name} n/a ({type(e).__name__}: {str(e)[:40]})")
    print(f"K {K} N {N} M {M} B {B} ln {ln} res {res_on} act {act} | " + " | ".join(row), flush=True)
hip_layers.DEBUG_FLAGS = 0
print("worst max-abs error of the new kernel:", worst)
sys.exit(0 if worst < 2e-4 else 1)
