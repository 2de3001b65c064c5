#!/usr/bin/env python3
"""Dispatch table of the AMP convs' three forms, as tools/gemm_sweep.py is for the GEMMs: one AMP
iteration c2(a2(c1(a1(x)))) + x (hierspeechpp_speechsynthesizer.py:380-384; c1 dilation d, c2 dilation 1) at every stage of
the Generator for B in {1, 2, 4, 8, 16, 32, 64} x T in {50, 200, 1000} frames, as
    direct   act + direct MFMA conv + act + direct MFMA conv                          (4 launches)
    fft      both convs in the frequency domain, activations fused into the forward transforms   (6 launches)
    pair     ... with the inverse of c1, a2 and the forward of c2 in one launch        (5 launches; where the LDS takes it)
    policy   what hierspeechpp_speechsynthesizer.amp_pair picks (fft_wins / fft_pair_ok)
hipGraph replay; a cell is flagged when the policy's time is more than 5 % above the best form's.
    python tools/fftconv_table.py > profiles/r05_fftconv_dispatch_table.txt"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import activations  # noqa: E402
from megatts2_hierspeechpp_amd import hierspeechpp_speechsynthesizer as hss  # noqa: E402
from megatts2_hierspeechpp_amd import _lib as L  # noqa: E402
from megatts2_hierspeechpp_amd.alias_free_torch import Activation1d  # noqa: E402
from megatts2_hierspeechpp_amd.hip_layers import Conv1d, finalize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, nargs="*", default=[1, 2, 4, 8, 16, 32, 64])
ap.add_argument("--frames", type=int, nargs="*", default=[50, 200, 1000])
ap.add_argument("--k", type=int, nargs="*", default=[11, 7])
ap.add_argument("--d", type=int, nargs="*", default=[1, 3, 5])
ap.add_argument("--max-elems", type=float, default=1.2e9, help="skip cells whose tensor has more elements (time, not memory)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * reps)


class Pair(torch.nn.Module):
    def __init__(self, C_, k, d):
        super().__init__()
        mk = lambda dd: Conv1d(C_, C_, k, dilation=dd, padding=(k - 1) * dd // 2, weight_norm=True)
        self.c1, self.c2 = mk(d), mk(1)
        self.a1 = Activation1d(activations.SnakeBeta(C_, alpha_logscale=True))
        self.a2 = Activation1d(activations.SnakeBeta(C_, alpha_logscale=True))
        for c in (self.c1, self.c2):
            c.weight_v.data.normal_()
            c.weight_g.data.uniform_(0.3, 0.7)
            c.enable_fft()


print("C  k d |  B    T      L | direct ms | fft ms | pair ms | policy: form ms | best | policy / best")
worst, flagged, cells = 1.0, [], 0
for C_, up in ((512, 4), (256, 20), (128, 80), (64, 160)):
    for k in a.k:
        for d in a.d:
            m = Pair(C_, k, d)
            finalize(m, dev)
            for T in a.frames:
                for B in a.batches:
                    Lx = up * T
                    if B * C_ * Lx > a.max_elems:
                        continue
                    x = torch.randn(B, C_, Lx, device=dev)
                    out = torch.empty_like(x)
                    run = lambda form: hss.amp_pair(m.c1, m.c2, m.a1, m.a2, x, form=form, res=x, out=out)
                    t = {"direct": timed(lambda: run("direct"))}
                    sup = m.c1.fft_supported(B, Lx) and m.c2.fft_supported(B, Lx)
                    if sup:
                        t["fft"] = timed(lambda: run("fft"))
                        if m.c1.fft_pair_ok(m.c2, x):
                            t["pair"] = timed(lambda: run("pair"))
                    # (the tool enables the form on every conv so that all columns exist; the model only where fft_eligible says so)
                    w1 = hss.fft_eligible(C_, k, d) and hss.fft_wins(m.c1, x)
                    w2 = hss.fft_eligible(C_, k, 1) and hss.fft_wins(m.c2, x)
                    pol = "direct" if not (w1 or w2) else ("pair" if w1 and w2 and hss.FFT_PAIR and hss.fft_act(x) and
                                                           m.c1.fft_pair_ok(m.c2, x) else ("fft" if w1 and w2 else "mixed"))
                    t_pol = t[pol] if pol in t else timed(lambda: run(None))
                    best = min(t, key=t.get)
                    ratio = t_pol / t[best]
                    cells += 1
                    worst = max(worst, ratio)
                    flag = "  <-- policy > 5 % off" if ratio > 1.05 else ""
                    if flag:
                        flagged.append((C_, k, d, B, T, pol, best, ratio))
                    f = lambda n: f"{t[n]:7.3f}" if n in t else "      -"
                    print(f"{C_:3d} {k:2d} {d} | {B:2d} {T:4d} {Lx:6d} | {f('direct')} | {f('fft')} | {f('pair')} | {pol:6s} {t_pol:7.3f} | "
                          f"{best:6s} | {ratio:5.2f}{flag}", flush=True)
                    del x, out
            del m
            torch.cuda.empty_cache()
print(f"# {cells} cells, policy within 5 % of the best form on {cells - len(flagged)}; worst ratio {worst:.2f}")
for f_ in flagged:
    print("# off:", f_)
