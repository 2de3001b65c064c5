#!/usr/bin/env python3
"""One modules.WN (H = 192, 8 layers, k = 5: the vocoder's posterior encoders) at T = 200: time per forward, one
hsp_wn_layer_f32 call per layer.  --debug words (tuning build only, HSP_LIB=.../libhsp_tune.so; results are then
wrong): 1 = producers stage the first two chunks only, 2 = consumers skip their MFMAs, 3 = both."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import hip_layers, modules  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--debug", nargs="+", type=int, default=[0])
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
wn = modules.WN(192, 5, 1, 8, gin_channels=256)
for p in wn.parameters():
    p.data.normal_(0, 0.05)
hip_layers.finalize(wn, dev)
x = torch.randn(a.batch, 192, a.frames, device=dev)
mask = torch.ones(a.batch, 1, a.frames, device=dev)
g = torch.randn(a.batch, 256, 1, device=dev)
for _ in range(3):
    wn(x, mask, g=g)
torch.cuda.synchronize()
for rep in range(2):            # twice: the first pass over the variants also warms the clocks
    for dbg in a.debug:
        hip_layers.DEBUG_FLAGS = dbg
        wn(x, mask, g=g)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()          # replayed: the eager loop is bound by Python (~100 us per layer)
        with torch.cuda.graph(graph):
            wn(x, mask, g=g)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"pass {rep} WN H192 x8 B {a.batch} T {a.frames} debug {dbg}: {e0.elapsed_time(e1) / a.reps * 1e3:9.1f} us per forward "
              f"({e0.elapsed_time(e1) / a.reps / 8 * 1e3:6.1f} us per layer incl. cond / mask launches)", flush=True)
