#!/usr/bin/env python3
"""Same-box A/B of the PLM loop (B = 16, T = 200, hipGraph replay): the loop as shipped (argmax folded into the next
step's embedding launch, last layer's residual read in place at column stride T) against the same loop with the two
extra launches per step of the earlier form (separate argmax, copy_strided of the last position)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megatts2_hierspeechpp_amd import _lib as L, functional as Fh, synth  # noqa: E402
from megatts2_hierspeechpp_amd.ttv_v1 import t2w2v_transformer as TT, transformer_mega as TM  # noqa: E402

dev = torch.device("cuda:0")
m = TT.Megatts2PLM1()
m.load_state_dict({k: torch.from_numpy(synth.synth_tensor("plm." + k, tuple(v.shape), 7)) for k, v in m.state_dict().items()})
m.finalize(dev)
tc = torch.from_numpy(np.random.default_rng(1).standard_normal((16, 256, 200)).astype(np.float32)).to(dev)


def capture():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m.infer(tc)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = m.infer(tc)
    return g, out


g_new, c_new = capture()

new_embed, new_fwd = TT.Megatts2PLM1._embed, TM.TransformerEncoderLayer.forward


def old_embed(self, tc_, codes, n, prev_logits=None):
    if prev_logits is not None:
        B = tc_.shape[0]
        L.check(L.lib().hsp_argmax_f32(L.fptr(prev_logits), 1, B, B, self.vq_bins, L.ptr(codes[:, n - 1:]), codes.stride(0),
                                       L.stream_ptr()), "hsp_argmax_f32")
    return new_embed(self, tc_, codes, n, None)


def old_fwd(self, x, mask=None, batch=None, last_only=False):
    res = x
    if last_only:
        B, T = batch
        res = Fh.copy_strided(x[0][:, :B * T].reshape(-1, B, T)[:, :, T - 1].unsqueeze(0))
    x = self.attn(x, mask=mask, res=res, batch=batch, last_only=last_only)
    h = self.ff["0"](x, act=L.ACT_RELU)
    return self.ff["3"](h, res=x)


TT.Megatts2PLM1._embed, TM.TransformerEncoderLayer.forward = old_embed, old_fwd
g_old, c_old = capture()
for g in (g_new, g_old):
    g.replay()
torch.cuda.synchronize()
print("codes equal:", bool((c_new == c_old).all()))
for r in range(3):
    for name, g in (("folded", g_new), ("two more launches per step", g_old)):
        t0 = time.perf_counter()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print(f"round {r} {name}: {1e3 * (time.perf_counter() - t0) / 3:.2f} ms", flush=True)
