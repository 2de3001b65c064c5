#!/usr/bin/env python3
"""Eager launches of the three transform kernels of the frequency-domain convs (dftseg_fwd_kernel with the fused activation,
dftseg_inv_kernel, dftseg_pair_kernel) at the Generator's stage shapes, B = 32 -- the workload of tools/pmc_dftseg.sh
(rocprofv3 --pmc wraps this script directly).  The channel products between them run too (they are what fills the spectrum).
    python tools/dftseg_eager.py [--reps 2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import activations  # noqa: E402
from megatts2_hierspeechpp_amd.alias_free_torch import Activation1d  # noqa: E402
from megatts2_hierspeechpp_amd.hip_layers import Conv1d, finalize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)


class Pair(torch.nn.Module):
    def __init__(self, C_, k, d):
        super().__init__()
        mk = lambda dd: Conv1d(C_, C_, k, dilation=dd, padding=(k - 1) * dd // 2, weight_norm=True)
        self.c1, self.c2 = mk(d), mk(1)
        self.a1 = Activation1d(activations.SnakeBeta(C_, alpha_logscale=True))
        self.a2 = Activation1d(activations.SnakeBeta(C_, alpha_logscale=True))
        for c in (self.c1, self.c2):
            c.weight_v.data.normal_()
            c.weight_g.data.uniform_(0.3, 0.7)
            c.enable_fft()


for C_, L_, k in ((512, 800, 11), (512, 800, 7), (256, 4000, 11), (256, 4000, 7), (128, 16000, 11), (128, 16000, 7), (64, 32000, 11)):
    for d in (1, 3, 5):
        m = Pair(C_, k, d)
        finalize(m, dev)
        x = torch.randn(a.batch, C_, L_, device=dev)
        out = torch.empty_like(x)
        for _ in range(a.reps):
            # three launches per conv: forward (activation fused), product, inverse (residual epilogue)
            xt = m.c1.forward_fft(x, act1d=m.a1)
            m.c2.forward_fft(xt, act1d=m.a2, res=x, out=out)
            if m.c1.fft_pair_ok(m.c2, x):
                m.c1.forward_fft_pair(m.c2, x, act_first=m.a1, act_second=m.a2, res=x, out=out)
        torch.cuda.synchronize()
        print(f"C {C_} L {L_} k {k} d {d}: pair {'yes' if m.c1.fft_pair_ok(m.c2, x) else 'no'}", flush=True)
        del m, x, out, xt
        torch.cuda.empty_cache()
