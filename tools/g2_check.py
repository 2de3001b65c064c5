#!/usr/bin/env python3
"""Fused WN layer / FFN (csrc/hsp_gemm2.hip) against the same layers launched one by one: error map by row block /
column (kernel bring-up)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L, hip_layers, modules  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 200


def unfused_group(kind, fn, structs, *extra):
    import ctypes as C
    for e in structs:
        if e is None:
            continue
        a = e[0]
        direct = a.Cin < 8 or a.Cout < 8 or a.Lout < 8
        L.check((L.lib().hsp_conv1d_direct_f32 if direct else L.lib().hsp_conv1d_mfma_f32)(C.byref(a), L.stream_ptr()), kind)


def run(module, *args, **kw):
    mod = module
    out_f = mod(*args, **kw)
    torch.cuda.synchronize()
    saved = hip_layers.launch_group
    hip_layers.launch_group = unfused_group
    try:
        out_u = mod(*args, **kw)
        torch.cuda.synchronize()
    finally:
        hip_layers.launch_group = saved
    return out_f, out_u


def report(name, f, u):
    d = (f - u).abs()
    print(f"{name}: max |fused - unfused| = {d.max().item():.3e}  (|ref| max {u.abs().max().item():.3e}); nan {torch.isnan(f).any().item()}")
    if d.max().item() > 1e-3:
        rows = d.amax(dim=(0, 2))
        cols = d.amax(dim=(0, 1))
        print("  bad 32-row blocks:", [i for i in range(rows.numel() // 32) if rows[32 * i:32 * i + 32].max() > 1e-3])
        print("  bad rows in block 0:", [i for i in range(32) if rows[i] > 1e-3])
        print("  bad columns:", [i for i in range(cols.numel()) if cols[i] > 1e-3][:40])


for nl, variant in ((1, "full"), (1, "xzero"), (1, "w2ident"), (2, "full")):
    wn = modules.WN(192, 5, 1, nl, gin_channels=256)
    for n_, p in wn.named_parameters():
        p.data.normal_(0, 0.05)
        if variant == "nobias" and n_.endswith("bias"):
            p.data.zero_()
        if variant == "w2ident" and "res_skip" in n_:
            # res_skip = identity: the output IS the activation tensor (weight_v = I, weight_g = 1, bias 0)
            if n_.endswith("weight_v"):
                p.data.copy_(torch.eye(192).reshape(192, 192, 1))
            elif n_.endswith("weight_g"):
                p.data.fill_(1.0)
            else:
                p.data.zero_()
    hip_layers.finalize(wn, dev)
    x = torch.randn(B, 192, T, device=dev)
    if variant == "xzero":
        x.zero_()
    lens = torch.randint(T // 2, T + 1, (B,))
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, None, :].to(dev)
    g = torch.randn(B, 256, 1, device=dev)
    f, u = run(wn, x, mask, g=g)
    report(f"WN {nl} layer(s) {variant} B {B} T {T}", f, u)

blk = modules.DiTConVBlock(192, 2, mlp_ratio=4.0, kernel=5)
for p in blk.parameters():
    p.data.normal_(0, 0.05)
hip_layers.finalize(blk, dev)
x = torch.randn(B, 192, T, device=dev) * mask
mod = torch.randn(B, 6 * 192, 1, device=dev) * 0.3
f, u = run(blk, x, None, mask, mod=mod, premasked=True)
report(f"DiT block B {B} T {T}", f, u)
