#!/usr/bin/env python3
"""BASELINE.json configs[3]: vocoder + SpeechSR48 super-resolution, batch 32 x 4 s on one MI355X.
    python tools/sr_bench.py [--batch 32] [--steps 5]
Prints one JSON line: 48 kHz output samples / s of the two-stage pipeline and the split."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megatts2_hierspeechpp_amd import synth  # noqa: E402
from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import SynthesizerTrn  # noqa: E402
from megatts2_hierspeechpp_amd.hip_layers import finalize  # noqa: E402
from megatts2_hierspeechpp_amd.speechsr48k.speechsr import SynthesizerTrn as SpeechSR  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3,
           p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
           upsample_rates=[4, 5, 4, 2, 2], upsample_initial_channel=1024, upsample_kernel_sizes=[8, 11, 8, 4, 4],
           gin_channels=256)
net = SynthesizerTrn(641, 192, **cfg)
sr = SpeechSR(128, 30, "0", [3, 7, 11], [[1, 3, 5]] * 3, [3], 32, [3])
holder = torch.nn.ModuleDict({"voc": net, "sr": sr})
holder.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in holder.state_dict().items()})
finalize(holder, dev)
B, T = args.batch, 200
inp = synth.synth_inputs(B, T, seed=1)
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}


def step():
    o, _ = net.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])
    return o, sr(o)


step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    o16, o48 = step()
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    g.replay()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.steps
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record(); o, _ = net.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"]); e[1].record(); sr(o); e[2].record()
torch.cuda.synchronize()
assert o48.shape == (B, 1, 3 * 320 * T) and bool(torch.isfinite(o48).all())
print(json.dumps({"metric": "48 kHz samples/s, vocoder + SpeechSR48, batch=32 (BASELINE.json configs[3])",
                  "value": B * 3 * 320 * T / el, "unit": "samples/s", "ms_per_step": el * 1e3,
                  "rtf": el / (B * 320 * T / 16000.0), "n_gpus": 1, "dtype": "f32", "data": "synthetic",
                  "stage_ms_eager": {"vocoder": e[0].elapsed_time(e[1]), "speechsr48": e[1].elapsed_time(e[2])}}))
