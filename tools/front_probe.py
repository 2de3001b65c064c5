#!/usr/bin/env python3
"""What would shortening the 50 Hz part's launch chains buy on the step?    Same-box, same process,
alternating hipGraph replays of the 32 x 4 s step:
    real            the product step
    no_ln1          every DiT block's FIRST LayerNorm launch removed (the identity in its place: wrong audio) -- an upper bound
                    for folding LayerNorm + modulate into the qkv GEMM (the fold keeps the statistics' arithmetic, only the
                    launch goes)
    no_ln           both LayerNorm launches of every block removed (192 launches)
    front_only / generator_only   the two halves of the step as their own graphs (what the front part costs alone)
The knobs live HERE (monkeypatches while the graph is captured); the product modules have no switch that turns an op into
the identity.
    python tools/front_probe.py [--rounds 3] [--json out.json]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from megatts2_hierspeechpp_amd import commons  # noqa: E402
from megatts2_hierspeechpp_amd import functional as Fh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--json", default=None)
a = ap.parse_args()
args = bench.parse_args(["--steps", str(a.steps)])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.VocoderWorkload(args, 0, 1, dev)
wl.model.finalize(dev)
wl.prepare(0, args.batch)
net, d = wl.model, wl.inp
_ln = Fh.layernorm_mod


def patched(skip_masked_only):
    def f(x, eps, mask=None, shift=None, scale=None, gamma=None, beta=None):
        if shift is not None and (mask is not None or not skip_masked_only):
            return x                                    # a DiT block's norm1 (mask given) / norm2: launch removed
        return _ln(x, eps, mask=mask, shift=shift, scale=scale, gamma=gamma, beta=beta)
    return f


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    return g, keep


steps = {}
steps["real"] = capture(wl.eager_step)
Fh.layernorm_mod = patched(True)
steps["no_ln1"] = capture(wl.eager_step)
Fh.layernorm_mod = patched(False)
steps["no_ln"] = capture(wl.eager_step)
Fh.layernorm_mod = _ln
x_mask = commons.sequence_mask(d["length"], d["mel"].size(2))
g_ = net.emb_g(d["mel"], x_mask).unsqueeze(-1)
z = net._latent(d["w2v"], d["f0"], x_mask, g_, d["noise"], 1.0)
steps["front_only"] = capture(lambda: net._latent(d["w2v"], d["f0"], x_mask, g_, d["noise"], 1.0))
steps["generator_only"] = capture(lambda: net._decode(z, g_))
out = {k: [] for k in steps}
for r in range(a.rounds):
    for name, (g, _) in steps.items():
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g.replay()
        torch.cuda.synchronize()
        out[name].append(1e3 * (time.perf_counter() - t0) / a.steps)
        print(f"round {r} {name:16s}: {out[name][-1]:.2f} ms", flush=True)
med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
print(json.dumps(med))
if a.json:
    with open(a.json, "w") as fh:
        json.dump({"ms": out, "median": med,
                   "workload": "bench.py VocoderWorkload 32 x 4 s, hipGraph replays; no_ln1 / no_ln: LayerNorm launches of the DiT "
                               "blocks replaced by the identity while capturing (wrong audio, tools only)"}, fh, indent=1)
