#!/usr/bin/env python3
"""Token-GEMM shapes (1x1 convs over a column axis) timed by hipGraph replay of 50 back-to-back launches, so that the
host's launch cost (3-15 us per Python call) is out of the number.  Variants = tuning words (libhsp_tune.so):
0 = dispatcher default, 131072 = the LDS-DMA token GEMM, 4 / 8 / 16 = register-path GEMM with 64x64 / 64x32 / 32x32 tiles.
    HSP_LIB=megatts2_hierspeechpp_amd/libhsp_tune.so python tools/gemm_bench.py --shapes K:N:M[:res[:B[:ln]]] ..."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402
from megatts2_hierspeechpp_amd.ttv_v1.transformer_mega import LayerNorm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="+", required=True)
ap.add_argument("--debug", nargs="+", type=int, default=[0, 131072, 4, 8, 16])
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for shp in a.shapes:
    f = [int(v) for v in shp.split(":")]
    K, N, M = f[:3]
    res_on = f[3] if len(f) > 3 else 0
    B = f[4] if len(f) > 4 else 1
    ln = f[5] if len(f) > 5 else 0

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
    x = torch.randn(B, K, N, device=dev)
    res = torch.randn(B, M, N, device=dev) if res_on else None
    out = torch.empty(B, M, N, device=dev)
    fl = 2.0 * B * K * M * N
    row = []
    for dbg in a.debug:
        hip_layers.DEBUG_FLAGS = dbg
        try:
            m.lin(x, res=res, out=out)
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(a.reps):
                        m.lin(x, res=res, out=out)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (4 * a.reps) * 1e3
            row.append(f"dbg{dbg}: {us:6.1f} us {fl / us / 1e6:5.1f} TF")
        except Exception as e:  # noqa: BLE001
            row.append(f"dbg{dbg}: n/a ({type(e).__name__})")
    print(f"K {K:4d} N {N:5d} M {M:4d} res {res_on} B {B:2d} ln {ln} | " + " | ".join(row), flush=True)
