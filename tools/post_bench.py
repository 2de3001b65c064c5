#!/usr/bin/env python3
"""activation_post -> conv_post -> tanh (two launches: the stand-alone activation, then the one-output-channel conv
kernel with the tanh epilogue) at the vocoder's (C = 32, L = 64 000) and SpeechSR48's (C = 32, L = 192 000) sizes, B = 32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L, activations, functional as Fh, hip_layers  # noqa: E402
from megatts2_hierspeechpp_amd.alias_free_torch import Activation1d  # noqa: E402

dev = torch.device("cuda:0")


class M(torch.nn.Module):
    def __init__(self, C_):
        super().__init__()
        self.activation_post = Activation1d(activations.SnakeBeta(C_, alpha_logscale=True))
        self.conv_post = hip_layers.Conv1d(C_, 1, 7, padding=3, bias=False)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for C_, Ls in ((32, 64000), (32, 192000)):
    m = M(C_)
    m.conv_post.weight.data.normal_(0, 0.1)
    hip_layers.finalize(m, dev)
    x = torch.randn(32, C_, Ls, device=dev)
    ax = m.activation_post(x)
    act_us = timeit(lambda: m.activation_post(x))
    conv_us = timeit(lambda: m.conv_post(ax, act=L.ACT_TANH))
    gb = 4 * 32 * (C_ + 1) * Ls / 1e9
    print(f"C {C_} L {Ls} B 32: activation {act_us:8.1f} us | conv_post + tanh {conv_us:8.1f} us "
          f"({gb / conv_us * 1e6:6.0f} GB/s of its algorithmic bytes)", flush=True)
