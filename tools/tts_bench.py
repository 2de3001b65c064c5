#!/usr/bin/env python3
"""BASELINE.json configs[2]: full text -> 16 kHz synthesis on one MI355X, batch 16: phone ids
U{12..112} [B, 40], tone U{0..10}, prompt mel [B, 80, 150], durations pinned to 10 frames / phone (-> 200 PLM steps, 4 s
of audio per utterance), synthetic weights.  The workload lives in tools/bench_extra.py (bench.py prints it as
extra_configs.tts_b16); this is its command line.

    python tools/tts_bench.py [--batch 16] [--phones 40] [--steps 3] [--warmup 1] [--no-graph]

Prints one JSON line with the whole-step rate and the per-stage split (HIP events)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import bench_extra  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--phones", type=int, default=40)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--no-graph", action="store_true", help="launch everything eagerly instead of replaying hipGraphs")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
print(json.dumps(bench_extra.tts_b16(dev, steps=args.steps, warmup=args.warmup, batch=args.batch, phones=args.phones,
                                     use_graph=not args.no_graph)))
