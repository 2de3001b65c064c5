#!/usr/bin/env python3
"""Throughput of the vocoder step when successive batches overlap (32 x 4 s each, hipGraph replay):
  a  one graph, replays back to back on one stream (the round-1..3 bench flow)
  b  two whole-step graphs replayed alternately on two streams, no cross dependencies
  c  step cut into F (style encoder + prior + flows + SourceNetwork: ~13 ms of short launches) and G (Generator):
     F of batch i+1 runs under G of batch i; G's serialised on their own stream, two buffer slots
Prints ms per batch for each."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from megatts2_hierspeechpp_amd import commons  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
args = bench.parse_args([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.VocoderWorkload(args, 0, 1, dev)
wl.model.finalize(dev)
wl.prepare(0, args.batch)
net, d = wl.model, wl.inp


def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def timed(submit, k=K, rounds=3):
    res = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        submit(k)
        torch.cuda.synchronize()
        res.append(1e3 * (time.perf_counter() - t0) / k)
    return res


with torch.no_grad():
    whole = lambda: net.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])
    whole()
    torch.cuda.synchronize()
    g0, o0 = capture(whole)
    g1, o1 = capture(whole)

    def front():
        x_mask = commons.sequence_mask(d["length"], d["mel"].size(2))
        g = net.emb_g(d["mel"], x_mask).unsqueeze(-1)
        z = net._latent(d["w2v"], d["f0"], x_mask, g, d["noise"], 1.0)
        e, e_ = net.sn(z, g)
        return z, g, e, e_

    F, G = [], []
    for slot in range(2):
        gf, fo = capture(front)
        gg, go = capture(lambda: net.dec(fo[0], fo[2], g=fo[1]))
        F.append((gf, fo))
        G.append((gg, go))
    torch.cuda.synchronize()
    for gr in (g0, F[0][0], G[0][0], F[1][0], G[1][0]):
        gr.replay()
    torch.cuda.synchronize()
    print("split vs whole step, max abs:", float((G[0][1] - o0[0]).abs().max()), float((G[1][1] - o0[0]).abs().max()))

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def sub_a(k):
    for _ in range(k):
        g0.replay()


def sub_b(k):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    for s in (s1, s2):
        s.wait_event(ev)
    for i in range(k):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            (g0 if i % 2 == 0 else g1).replay()
    for s in (s1, s2):
        e = torch.cuda.Event()
        e.record(s)
        main.wait_event(e)


def sub_c(k):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    for s in (s1, s2):
        s.wait_event(ev)
    g_done = [None, None]
    for i in range(k):
        slot = i % 2
        with torch.cuda.stream(s1):
            if g_done[slot] is not None:
                s1.wait_event(g_done[slot])      # the slot's buffers are free once its last Generator pass is done
            F[slot][0].replay()
            fe = torch.cuda.Event()
            fe.record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(fe)
            G[slot][0].replay()
            ge = torch.cuda.Event()
            ge.record(s2)
            g_done[slot] = ge
    for s in (s1, s2):
        e = torch.cuda.Event()
        e.record(s)
        main.wait_event(e)


for name, fn in (("a one graph back to back", sub_a), ("b two whole-step graphs, two streams", sub_b),
                 ("c F under G, two slots", sub_c), ("a again", sub_a), ("c again", sub_c)):
    fn(4)
    torch.cuda.synchronize()
    print(f"{name}: " + " ".join(f"{x:.2f}" for x in timed(fn)) + " ms per batch", flush=True)
