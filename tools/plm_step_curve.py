#!/usr/bin/env python3
"""Per-step time of the greedy PLM loop against the step index t ("the floor x 200 vs flops split on
file").  Step t re-encodes the whole prefix of t + 1 positions for B utterances (ttv_v1/t2w2v_transformer.py:702-718), so
its token count grows linearly and its attention quadratically; a HIP-event pair brackets every step of an eager loop (the
host runs ahead of the GPU: ~22 launches of 5 us against ~350 us of GPU time per step).  Prints the curve, a least-squares
fit  ms(t) = floor + a (t + 1) + b (t + 1)^2  and what each term sums to over the 200 steps.
    python tools/plm_step_curve.py [--batch 16 32] > profiles/r05_plm_step_curve.txt"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megatts2_hierspeechpp_amd import synth  # noqa: E402
from megatts2_hierspeechpp_amd.ttv_v1.t2w2v_transformer import Megatts2PLM1  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="*", default=[16])
ap.add_argument("--frames", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
m = Megatts2PLM1()
m.load_state_dict({k: torch.from_numpy(synth.synth_tensor("plm." + k, tuple(v.shape), 7)) for k, v in m.state_dict().items()})
m.finalize(dev)
for B in a.batch:
    T = a.frames
    tc = torch.from_numpy(np.random.default_rng(1).standard_normal((B, 256, T)).astype(np.float32)).to(dev)
    m.infer(tc)
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):
        codes = torch.empty(B, T + 1, dtype=torch.int64, device=dev)
        codes[:, 0] = m.GO_ID
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(T + 1)]
        lg = None
        ev[0].record()
        for t in range(T):
            lg = m.step_logits(tc, codes, t + 1, prev_logits=lg)
            ev[t + 1].record()
        torch.cuda.synchronize()
        runs.append([ev[t].elapsed_time(ev[t + 1]) for t in range(T)])
    ms = np.median(np.array(runs), axis=0)
    n = np.arange(1, T + 1, dtype=np.float64)
    A = np.stack([np.ones_like(n), n, n * n], axis=1)
    (c0, c1, c2), *_ = np.linalg.lstsq(A, ms, rcond=None)
    print(f"# B = {B}, T = {T}: eager loop {ms.sum():.1f} ms (sum of the per-step medians of 3 runs)")
    print(f"# fit ms(t) = {c0 * 1e3:.1f} us + {c1 * 1e3:.3f} us x (t + 1) + {c2 * 1e6:.3f} ns x (t + 1)^2   "
          f"-> floor {c0 * T:.1f} ms, linear (tokens) {c1 * n.sum():.1f} ms, quadratic (attention) {c2 * (n * n).sum():.1f} ms")
    print("t  tokens  ms_per_step")
    for t in list(range(0, 10)) + list(range(10, T, 10)) + [T - 1]:
        print(f"{t:3d} {B * (t + 1):6d} {ms[t]:.4f}")
