#!/usr/bin/env python3
"""In-kernel phase timeline of the fused WN layer (tuning build: HSP_LIB=.../libhsp_tune.so): s_memtime stamps of
the first consumer / producer wave of every workgroup, averaged.  Stamps: 0 start, 1 tables filled, 2 prologue issued,
3 phase 1 done, 4 gate done, 5 epilogue operands requested, 6 phase 2 done, 7 stores issued."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L, hip_layers, modules  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T = 32, 200
wn = modules.WN(192, 5, 1, 2, gin_channels=256)
for p in wn.parameters():
    p.data.normal_(0, 0.05)
hip_layers.finalize(wn, dev)
x = torch.randn(B, 192, T, device=dev)
mask = torch.ones(B, 1, T, device=dev)
g = torch.randn(B, 256, 1, device=dev)
nblk = B * ((T + 31) // 32)
buf = torch.zeros(nblk * 2 * 8, dtype=torch.int64, device=dev)
orig = hip_layers.Conv1d.forward


def fwd(self, x_, **kw):   # route the stamp buffer through the (otherwise unused) filt field of the gated conv
    out = orig(self, x_, **kw)
    if hip_layers._DEFER and self.rows == L.ROWS_GATE_WN:
        hip_layers._DEFER[-1][0].filt = buf.data_ptr()
    return out


hip_layers.Conv1d.forward = fwd
for dbg in [int(v) for v in sys.argv[1:]] or [256]:
    hip_layers.DEBUG_FLAGS = dbg
    for _ in range(5):
        wn(x, mask, g=g)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(nblk, 2, 8).astype(np.float64)
    for role, name in ((0, "consumer wave 0"), (1, "producer wave 0")):
        d = st[:, role, :] - st[:, role, :1]
        print(f"debug {dbg} {name}: mean ticks since start (100 MHz: x10 ns) " + " ".join(f"{v:9.0f}" for v in d.mean(0)),
              " | max end", d[:, 7].max() if role == 0 else d[:, 6].max())
    span = st[:, 0, 7].max() - st[:, 0, 0].min()
    print(f"   first start -> last end over the grid: {span:.0f} ticks; start spread {st[:, 0, 0].max() - st[:, 0, 0].min():.0f}")
