#!/usr/bin/env python3
"""Time hsp_mha_f32 at the PLM shapes (B x H x D, the batch side by side on the columns) in a hipGraph of 50 calls.
    python tools/mha_bench.py [--batch 16] [--heads 4] [--dim 69]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import functional as Fh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--heads", type=int, default=4)
ap.add_argument("--dim", type=int, default=69)
ap.add_argument("--lens", default="16,32,64,100,128,160,200")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, H, D = a.batch, a.heads, a.dim
for T in [int(t) for t in a.lens.split(",")]:
    qkv = torch.randn(3 * H * D, B * T, device=dev)
    o = torch.empty(H * D, B * T, device=dev)
    per = lambda m: m.reshape(-1, B, T).permute(1, 0, 2)
    q, k, v = (per(qkv[i * H * D:(i + 1) * H * D]) for i in range(3))
    run = lambda: Fh.mha(q, k, v, H, D ** -0.5, out=per(o))
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    fl = 4.0 * B * H * T * T * D
    print(f"T={T:4d}: {us:7.1f} us / call   {fl / us / 1e6:6.2f} TFLOP/s (algorithmic)  blocks={B * H * ((T + 31) // 32)}")
