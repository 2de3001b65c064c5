import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from tools import bench_extra as BE
from megatts2_hierspeechpp_amd import inference_plm as IP, synth
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
B, N = 16, 40
models = IP.TtsModels(BE.VOC_CFG, BE.TTV_CFG)
models.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in models.state_dict().items()})
models.finalize(dev)
r = np.random.default_rng(3)
ids = torch.from_numpy(r.integers(12, 113, (B, N))).to(dev); tone = torch.from_numpy(r.integers(0, 11, (B, N))).to(dev)
lang = torch.where(ids < 74, 1, 2).to(dev); tlen = torch.full((B,), N, dtype=torch.int64, device=dev)
mel = torch.from_numpy(synth.synth_inputs(B, 150, seed=5)["mel"]).to(dev); mlen = torch.full((B,), 150, dtype=torch.int64, device=dev)
mel2, mlen2 = torch.cat([mel, mel]), torch.cat([mlen, mlen]); dur = torch.full((B, N), 10.0, device=dev)
T2 = 200; noise = torch.from_numpy(r.standard_normal((B, 192, T2)).astype(np.float32)).to(dev)
x_frame, g, x_lengths, x_mask = models.ttv.inf_extract_tc_latent(ids, tlen, mel, mlen, tone, lang, dur=dur)
def back(codes):
    w2v, pitch = models.ttv.inf_plm_gen(x_frame, g, codes, x_lengths, x_mask)
    pitch = IP.zero_below(pitch, float(np.log(55.0)))
    frames = torch.ceil(x_lengths).to(torch.int64)
    audio = models.voc.voice_conversion_noise_control(w2v, frames, mel2, mlen2, pitch.unsqueeze(1), noise_scale=0.333, denoise_ratio=0.0, noise=noise)
    return IP.peak_int16(audio, frames * 320)
codes = models.plm.infer(x_frame); back(codes); torch.cuda.synchronize()
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1): c1 = models.plm.infer(x_frame)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2): w2 = back(c1)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    c3 = models.plm.infer(x_frame); w3 = back(c3)
print("plm graph", round(t(g1.replay), 1), "| back eager", round(t(lambda: back(c1)), 1), "| back graph", round(t(g2.replay), 1),
      "| plm graph + back eager", round(t(lambda: (g1.replay(), back(c1))), 1), "| two graphs", round(t(lambda: (g1.replay(), g2.replay())), 1),
      "| one graph", round(t(g3.replay), 1))
def seq_sync():
    g1.replay(); torch.cuda.synchronize(); g2.replay()
print("plm graph, sync, back graph", round(t(seq_sync), 1), "| plm graph, sync, back eager", round(t(lambda: (g1.replay(), torch.cuda.synchronize(), back(c1))), 1))
