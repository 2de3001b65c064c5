#!/usr/bin/env python3
"""Wall time of the vocoder step's parts, each captured into its own hipGraph (32 x 4 s, product launch policy):
front = style encoder + SF prior encoder + two reversed flows (50 Hz); sn = SourceNetwork; dec = Generator."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from megatts2_hierspeechpp_amd import commons  # noqa: E402

args = bench.parse_args([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.VocoderWorkload(args, 0, 1, dev)
wl.model.finalize(dev)
wl.prepare(0, args.batch)
net, d = wl.model, wl.inp


def graph_ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out


with torch.no_grad():
    x_mask = commons.sequence_mask(d["length"], d["mel"].size(2))
    ms_g, g = graph_ms(lambda: net.emb_g(d["mel"], x_mask).unsqueeze(-1))
    ms_f, z = graph_ms(lambda: net._latent(d["w2v"], d["f0"], x_mask, g, d["noise"], 1.0))
    ms_s, (e, e_) = graph_ms(lambda: net.sn(z, g))
    ms_d, o = graph_ms(lambda: net.dec(z, e, g=g))
    ms_all, _ = graph_ms(lambda: net.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"]))
print(f"style encoder {ms_g:.2f} | prior + flows {ms_f:.2f} | source network {ms_s:.2f} | generator {ms_d:.2f} | "
      f"sum {ms_g + ms_f + ms_s + ms_d:.2f} | whole step {ms_all:.2f} ms")
