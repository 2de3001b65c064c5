#!/usr/bin/env python3
"""debug: gated conv through the K-split tile (S64G2) against the 64 x 128 tile (tuning bit 1 << 27) and torch"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import hip_layers, _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (cin, H, k, T, B) in ((192, 192, 5, 50, 1), (192, 192, 5, 200, 2), (64, 32, 3, 40, 1), (32, 32, 1, 64, 1)):
    conv = hip_layers.Conv1d(cin, 2 * H, k, padding=(k - 1) // 2, rows=L.ROWS_GATE_WN)
    conv.weight.data.normal_(0, 0.05)
    conv.bias.data.normal_(0, 0.1)
    hip_layers.finalize(conv, dev)
    x = torch.randn(B, cin, T, device=dev)
    w, b = conv.weight.data.double().cpu(), conv.bias.data.double().cpu()
    y = torch.nn.functional.conv1d(x.double().cpu(), w, b, padding=(k - 1) // 2)
    ref = (torch.tanh(y[:, :H]) * torch.sigmoid(y[:, H:])).float()
    out = {}
    for dbg in (0, 1 << 27):
        hip_layers.DEBUG_FLAGS = dbg
        out[dbg] = conv(x).cpu()
    hip_layers.DEBUG_FLAGS = 0
    e_new, e_old = (out[0] - ref).abs(), (out[1 << 27] - ref).abs()
    print(f"cin {cin} H {H} k {k} T {T} B {B}: new {float(e_new.max()):.3e} old {float(e_old.max()):.3e}")
    if float(e_new.max()) > 1e-3:
        bad = (e_new > 1e-3)
        rows = bad.any(dim=2)[0].nonzero().flatten().tolist()
        cols = bad.any(dim=1)[0].nonzero().flatten().tolist()
        print("  bad rows", rows[:40], "... n", len(rows), "| bad cols", cols[:40], "n", len(cols))
