#!/usr/bin/env python3
"""Run ONE of bench.py's extra configs alone and print its record (tools/bench_extra.py):
    python tools/extra_run.py vc_b1_4s | tts_prompt_denoise | tts_b16 | tts_b1 | tts_2x16 | sr48_b32 | vocoder_b1_1s [--steps N]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import bench_extra  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("--steps", type=int, default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
fn = getattr(bench_extra, a.name)
out = fn(dev, steps=a.steps) if a.steps else fn(dev)
print(json.dumps(out, indent=1))
