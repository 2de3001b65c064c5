#!/usr/bin/env python3
"""Time Megatts2PLM1.infer on cuda:0: eager launches vs one hipGraph replay.
python tools/plm_bench.py [--batch 16] [--frames 200] [--reps 3]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megatts2_hierspeechpp_amd import synth  # noqa: E402
from megatts2_hierspeechpp_amd.ttv_v1.t2w2v_transformer import Megatts2PLM1  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--debug", type=int, default=0, help="hsp_conv1d_args.debug for every conv launch (timing floors only)")
args = ap.parse_args()
from megatts2_hierspeechpp_amd import hip_layers  # noqa: E402
hip_layers.DEBUG_FLAGS = args.debug
dev = torch.device("cuda:0")
m = Megatts2PLM1()
m.load_state_dict({k: torch.from_numpy(synth.synth_tensor("plm." + k, tuple(v.shape), 7)) for k, v in m.state_dict().items()})
m.finalize(dev)
tc = torch.from_numpy(np.random.default_rng(1).standard_normal((args.batch, 256, args.frames)).astype(np.float32)).to(dev)
codes = m.infer(tc)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(args.reps):
    codes = m.infer(tc)
torch.cuda.synchronize()
eager = (time.time() - t0) / args.reps
print(f"eager: {eager * 1e3:.1f} ms / call  (B={args.batch}, T={args.frames})")
if not args.no_graph:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m.infer(tc)
        g = torch.cuda.CUDAGraph()
        t0 = time.time()
        with torch.cuda.graph(g, stream=s):
            gcodes = m.infer(tc)
        print(f"capture: {time.time() - t0:.1f} s")
    g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.reps):
        g.replay()
    torch.cuda.synchronize()
    gr = (time.time() - t0) / args.reps
    print(f"graph: {gr * 1e3:.1f} ms / call; codes equal eager: {bool((gcodes == codes).all())}")
    flops = 2 * args.batch * sum(t + 1 for t in range(args.frames)) * 4 * (4 * 276 * 276 + 2 * 276 * 1104)
    print(f"reference-algorithm GEMM flops {flops / 1e12:.2f} TFLOP -> {flops / gr / 1e12:.1f} TFLOP/s")
