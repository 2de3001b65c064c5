#!/usr/bin/env python3
"""Micro-benchmark of one fused conv shape (for kernel tuning / PMC profiling).
   python tools/conv_bench.py --cin 512 --cout 512 --k 11 --dil 1 --len 800 --batch 32 --act 1 --reps 20"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import activations, hip_layers  # noqa: E402
from megatts2_hierspeechpp_amd.alias_free_torch import Activation1d  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=512)
ap.add_argument("--cout", type=int, default=512)
ap.add_argument("--k", type=int, default=11)
ap.add_argument("--dil", type=int, default=1)
ap.add_argument("--len", type=int, default=800)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--act", type=int, default=1)
ap.add_argument("--res", type=int, default=0)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--debug", type=int, default=0)
ap.add_argument("--actonly", type=int, default=0, help="time the stand-alone activation kernel instead")
a = ap.parse_args()
hip_layers.DEBUG_FLAGS = a.debug
dev = torch.device("cuda:0")
torch.manual_seed(0)


class M(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = hip_layers.Conv1d(a.cin, a.cout, a.k, dilation=a.dil, padding=(a.k - 1) * a.dil // 2)
        self.act = Activation1d(activations.SnakeBeta(a.cin, alpha_logscale=True))


m = M()
m.conv.weight.data.normal_(0, 0.05)
m.conv.bias.data.normal_(0, 0.1)
m.act.act.alpha.data.normal_(0, 0.5)
m.act.act.beta.data.normal_(0, 0.5)
hip_layers.finalize(m, dev)
x = torch.randn(a.batch, a.cin, a.len, device=dev)
res = torch.randn(a.batch, a.cout, a.len, device=dev) if a.res else None
out = torch.empty(a.batch, a.cout, a.len, device=dev)
run = (lambda: m.act(x)) if a.actonly else (lambda: m.conv(x, act1d=m.act if a.act else None, res=res, out=out))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
fl = 2.0 * a.batch * a.cout * a.cin * a.k * a.len
if a.actonly:
    gb = 8.0 * a.batch * a.cin * a.len / 1e9
    print(f"act1d C {a.cin} L {a.len} B {a.batch}: {ms*1e3:9.1f} us  {gb/ms*1e3:7.1f} GB/s")
    sys.exit(0)
print(f"cin {a.cin} cout {a.cout} k {a.k} dil {a.dil} L {a.len} B {a.batch} act {a.act} dbg {a.debug}: {ms*1e3:9.1f} us  {fl/ms/1e9:7.2f} TF/s")
