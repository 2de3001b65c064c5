#!/usr/bin/env python3
"""Eager 1 x 1 s vocoder steps for `rocprofv3 --kernel-trace --stats` (where does a small request's latency go?):
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1 -- python3 $R/tools/b1_profile.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megatts2_hierspeechpp_amd import synth  # noqa: E402
from megatts2_hierspeechpp_amd import hierspeechpp_speechsynthesizer as hss  # noqa: E402
from tools.bench_extra import VOC_CFG  # noqa: E402

hss.AMP_STREAMS, hss.FRONT_SPLITS = 0, 1
dev = torch.device("cuda:0")
net = hss.SynthesizerTrn(641, 192, **VOC_CFG)
net.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in net.state_dict().items()})
net.finalize(dev)
d = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_inputs(1, 50, seed=2).items()}
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    net.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])
torch.cuda.synchronize()
