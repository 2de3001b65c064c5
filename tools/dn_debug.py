#!/usr/bin/env python3
"""Stage-by-stage comparison of the denoiser's HIP path with the CPU oracle on a golden input (debugging aid; run on
the GPU box: python tools/dn_debug.py [fixture])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import hsp_oracle as O  # noqa: E402
from megatts2_hierspeechpp_amd.denoiser import infer as DI  # noqa: E402
from megatts2_hierspeechpp_amd.hip_layers import finalize  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "denoise_l8000"
meta, arrays = H.load_fixture(name)
dev = torch.device("cuda:0")
mod = H.build_module(meta)
sd = H.synth_sd(meta)
mod.load_state_dict(sd, strict=True)
finalize(mod, dev)
osd = H.oracle_sd(meta)
wav = torch.from_numpy(arrays["wav"])
err = lambda a, b: float((a.detach().cpu() - b).abs().max())

norm = torch.sqrt(len(wav) / torch.sum(wav ** 2.0))
y = (wav * norm).unsqueeze(0)
win = torch.hann_window(400)
spec = torch.stft(y, 400, hop_length=100, win_length=400, window=win, center=True, pad_mode="reflect", return_complex=True)
mag_r, pha_r = torch.abs(spec) ** 0.3, torch.angle(spec)
mag, pha, _ = DI.mag_pha_stft(y.to(dev), 400, 100, 400, 0.3)
print("stft mag", err(mag, mag_r), "pha", err(pha, pha_r), "shape", tuple(mag.shape))
# feed the ORACLE's spectrogram to both sides from here on
x_r = torch.cat((mag_r.unsqueeze(-1).permute(0, 3, 2, 1), pha_r.unsqueeze(-1).permute(0, 3, 2, 1)), 1)
x = x_r.to(dev).contiguous()
h_r = O.mp_dense_encoder(osd, "dense_encoder", x_r)
h = mod.dense_encoder(x)
print("dense_encoder", err(h, h_r), "ref max", float(h_r.abs().max()))
for i, blk in enumerate(mod.TSConformer):
    h_r = O.mp_ts_conformer(osd, f"TSConformer.{i}", h_r)
    h = blk(h_r.to(dev).contiguous() if os.environ.get("RESYNC") else h)
    print(f"tsconformer {i}", err(h, h_r), "ref max", float(h_r.abs().max()))
amp_r, pha_g_r, _ = O.mpnet(osd, "", mag_r, pha_r)
amp, pha_g, _ = mod(mag_r.to(dev), pha_r.to(dev))
print("mpnet amp", err(amp, amp_r), "pha", err(pha_g, pha_g_r), "amp max", float(amp_r.abs().max()))
m = torch.pow(amp_r, 1.0 / 0.3)
wav_r = torch.istft(torch.complex(m * torch.cos(pha_g_r), m * torch.sin(pha_g_r)), 400, hop_length=100, win_length=400,
                    window=win, center=True)
wav_g = DI.mag_pha_istft(amp_r.to(dev), pha_g_r.to(dev), 400, 100, 400, 0.3)
print("istft", err(wav_g, wav_r), "ref max", float(wav_r.abs().max()))
d = (wav_g.cpu() - wav_r).abs()[0]
print("istft worst index", int(d.argmax()), "of", d.numel())
# ---- end to end on the oracle's spectrogram: which stage loses the golden audio?
gold = torch.from_numpy(arrays["out0"])
mg = torch.pow(amp.cpu(), 1.0 / 0.3)
wav_a = torch.istft(torch.complex(mg * torch.cos(pha_g.cpu()), mg * torch.sin(pha_g.cpu())), 400, hop_length=100,
                    win_length=400, window=win, center=True) / norm
wav_b = DI.mag_pha_istft(amp, pha_g, 400, 100, 400, 0.3, scale=1.0 / float(norm))
print("golden vs torch.istft(gpu amp, pha)", err(wav_a, gold), " vs hip istft(gpu amp, pha)", err(wav_b, gold),
      " vs oracle audio", err(wav_r / norm, gold))
dp = (pha_g.cpu() - pha_g_r)
print("pha diff max", float(dp.abs().max()), "amp^(1/c) max", float(mg.max()), "amp rel err", float(((amp.cpu() - amp_r).abs() / amp_r.abs().clamp_min(1e-3)).max()))
big = torch.pow(amp_r, 1 / 0.3)
print("decompressed magnitude: max", float(big.max()), "err", float((mg - big).abs().max()))
