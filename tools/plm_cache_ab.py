#!/usr/bin/env python3
"""(round 6) The PLM greedy loop with and without the layer-0 q / k / v cache (HSP_PLM_CACHE_L0), same process, hipGraph
replays alternating; codes of both forms compared.
    python tools/plm_cache_ab.py [--batch 16] [--T 200] [--rounds 3]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megatts2_hierspeechpp_amd import synth  # noqa: E402
from megatts2_hierspeechpp_amd.ttv_v1 import t2w2v_transformer as T2  # noqa: E402
from tools.bench_extra import event_median_ms  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--T", type=int, default=200)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--json", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
plm = T2.Megatts2PLM1()
plm.load_state_dict({k: torch.from_numpy(synth.synth_tensor("plm." + k, tuple(v.shape), 0)) for k, v in plm.state_dict().items()})
plm.finalize(dev)
tc = torch.from_numpy(np.random.default_rng(1).standard_normal((a.batch, 256, a.T)).astype(np.float32)).to(dev)
graphs, codes = {}, {}
for name, on in (("full re-projection", False), ("layer-0 cache", True)):
    T2.PLM_CACHE_L0 = on
    plm.infer(tc)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        codes[name] = plm.infer(tc)
    g.replay()
    torch.cuda.synchronize()
    graphs[name] = g
same = bool(torch.equal(codes["full re-projection"], codes["layer-0 cache"]))
print("codes equal:", same, "first difference at",
      None if same else (codes["full re-projection"] != codes["layer-0 cache"]).nonzero()[0].tolist())
out = {k: [] for k in graphs}
for r in range(a.rounds):
    for name, g in graphs.items():
        out[name].append(event_median_ms(g.replay, 5))
        print(f"round {r} {name:20s}: {out[name][-1]:.2f} ms per {a.batch} x {a.T} loop", flush=True)
if a.json:
    json.dump({"ms_per_loop": out, "codes_equal": same, "batch": a.batch, "T": a.T}, open(a.json, "w"), indent=1)
