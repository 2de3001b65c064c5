#!/usr/bin/env python3
"""BASELINE.json configs[2]: full text -> 16 kHz synthesis on one MI355X, batch 16 (SURVEY.md §8d
config 3): phone ids U{12..112} [B, 40], tone U{0..10}, prompt mel [B, 80, 150], durations pinned
to 10 frames / phone (-> 200 PLM steps, 4 s of audio per utterance), synthetic weights.

    python tools/tts_bench.py [--batch 16] [--phones 40] [--steps 3] [--warmup 1]

Prints one JSON line with the whole-step rate and the per-stage split (HIP events)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megatts2_hierspeechpp_amd import inference_plm as IP, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--phones", type=int, default=40)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--no-graph", action="store_true", help="launch the PLM loop eagerly instead of replaying its hipGraph")
args = ap.parse_args()
dev = torch.device("cuda:0")
B, N = args.batch, args.phones

voc_cfg = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3,
               p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
               upsample_rates=[4, 5, 4, 2, 2], upsample_initial_channel=1024, upsample_kernel_sizes=[8, 11, 8, 4, 4],
               gin_channels=256)
ttv_cfg = dict(inter_channels=256, hidden_channels=256, filter_channels=1024, n_heads=4, n_layers=6, kernel_size=3,
               p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
               use_spectral_norm=False)
models = IP.TtsModels(voc_cfg, ttv_cfg)
models.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in models.state_dict().items()})
models.finalize(dev)

r = np.random.default_rng(3)
ids = torch.from_numpy(r.integers(12, 113, (B, N))).to(dev)
tone = torch.from_numpy(r.integers(0, 11, (B, N))).to(dev)
lang = torch.where(ids < 74, 1, 2).to(dev)
tlen = torch.full((B,), N, dtype=torch.int64, device=dev)
mel = torch.from_numpy(synth.synth_inputs(B, 150, seed=5)["mel"]).to(dev)
mlen = torch.full((B,), 150, dtype=torch.int64, device=dev)
mel2, mlen2 = torch.cat([mel, mel]), torch.cat([mlen, mlen])
dur = torch.full((B, N), 10.0, device=dev)
T2 = N * 10 // 2
noise = torch.from_numpy(r.standard_normal((B, 192, T2)).astype(np.float32)).to(dev)

plm_graph = {}


def plm_infer(x_frame):
    if args.no_graph:
        return models.plm.infer(x_frame)
    if "g" not in plm_graph:
        plm_graph["x"] = x_frame.clone()
        models.plm.infer(plm_graph["x"])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            plm_graph["codes"] = models.plm.infer(plm_graph["x"])
        plm_graph["g"] = g
    plm_graph["x"].copy_(x_frame)
    plm_graph["g"].replay()
    return plm_graph["codes"]


def step(ev=None):
    mark = (lambda: ev.append(torch.cuda.Event(enable_timing=True)) or ev[-1].record()) if ev is not None else (lambda: None)
    mark()
    x_frame, g, x_lengths, x_mask = models.ttv.inf_extract_tc_latent(ids, tlen, mel, mlen, tone, lang, dur=dur)
    mark()
    codes = plm_infer(x_frame)
    mark()
    w2v, pitch = models.ttv.inf_plm_gen(x_frame, g, codes, x_lengths, x_mask)
    pitch = IP.zero_below(pitch, float(np.log(55.0)))
    mark()
    frames = torch.ceil(x_lengths).to(torch.int64)
    audio = models.voc.voice_conversion_noise_control(w2v, frames, mel2, mlen2, pitch.unsqueeze(1), noise_scale=0.333,
                                                      denoise_ratio=0.0, noise=noise)
    mark()
    wav = IP.peak_int16(audio, frames * 320)
    mark()
    return wav


for _ in range(max(args.warmup, 1)):
    wav = step()
torch.cuda.synchronize()
assert wav.shape == (B, 320 * T2), wav.shape
t0 = time.perf_counter()
for _ in range(args.steps):
    wav = step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.steps
ev = []
step(ev)
torch.cuda.synchronize()
names = ["front_end(A16-A17)", "plm_loop(A18)", "w2v+pitch(A17)", "vocoder(A1-A14)", "int16_post(A19)"]
stages = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
print(json.dumps({
    "metric": "16kHz audio samples/sec, full inference_plm.py text->wav, batch=16 (BASELINE.json configs[2])",
    "value": B * 320 * T2 / el, "unit": "samples/s", "ms_per_step": el * 1e3, "rtf": el / (B * 320 * T2 / 16000.0),
    "n_gpus": 1, "dtype": "f32", "data": "synthetic",
    "config": {"workload": f"tts: {B} utterances x {N} phones x 10 frames -> {320 * T2 / 16000:g} s each, prompt mel 150 frames",
               "plm_steps": T2, "plm_launch_mode": "eager" if args.no_graph else "hipGraph"},
    "stage_ms": stages}))
