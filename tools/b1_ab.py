#!/usr/bin/env python3
"""Latency of single requests (vocoder 1 x 1 s, 1 x 4 s, 4 x 4 s) under tuning words of the conv dispatcher, alternating
(tuning build: HSP_LIB=.../libhsp_tune.so).   python tools/b1_ab.py --debug 0 134217728"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import bench_extra as BE  # noqa: E402
from megatts2_hierspeechpp_amd import hip_layers, synth  # noqa: E402
from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import SynthesizerTrn  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--debug", type=int, nargs="+", default=[0])
a = ap.parse_args()
dev = torch.device("cuda:0")
net = SynthesizerTrn(641, 192, **BE.VOC_CFG)
net.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in net.state_dict().items()})
net.finalize(dev)
for rnd in range(2):
    for frames in (50, 200):
        for dbg in a.debug:
            hip_layers.DEBUG_FLAGS = dbg
            r = BE.vocoder_b1_1s(dev, steps=30, net=net, frames=frames)
            print(f"round {rnd} 1 x {frames / 50:g} s debug {dbg}: {r['ms_per_step']:.3f} ms", flush=True)
hip_layers.DEBUG_FLAGS = 0
