#!/usr/bin/env python3
"""Two TTS batches in flight: the greedy PLM loop of batch i + 1 on a stream created with
hipExtStreamCreateWithCUMask beside the back half (w2v / pitch decoder + vocoder + int16) of batch i on the complement.
The loop is 200 dependent steps of ~22 small launches -- launch floors, most CUs idle -- while the vocoder is MFMA-bound:

  1. does a captured hipGraph keep the launch stream's CU mask?   (a full-chip conv graph on an N-CU stream: time x 256 / N if so)
  2. the PLM graph alone on N CUs, the back-half graph alone on the complement
  3. the pipeline: per iteration [PLM graph of batch i + 1 on the masked stream] || [front-end of batch i + 2, back-half graph of
     batch i on the complement], one host synchronisation per iteration; against the sequential flow of bench_extra.tts_b16

    python tools/plm_overlap_probe.py [--cus 16 32 64] [--json out.json]"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import bench_extra as BE  # noqa: E402
from megatts2_hierspeechpp_amd import inference_plm as IP, synth  # noqa: E402
from megatts2_hierspeechpp_amd import parallel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cus", type=int, nargs="*", default=[16, 32, 64])
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--json", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
B, N, T2 = 16, 40, 200
NCU = torch.cuda.get_device_properties(dev).multi_processor_count

models = IP.TtsModels(BE.VOC_CFG, BE.TTV_CFG)
models.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in models.state_dict().items()})
models.finalize(dev)
r = np.random.default_rng(3)
ids = torch.from_numpy(r.integers(12, 113, (B, N))).to(dev)
tone = torch.from_numpy(r.integers(0, 11, (B, N))).to(dev)
lang = torch.where(ids < 74, 1, 2).to(dev)
tlen = torch.full((B,), N, dtype=torch.int64, device=dev)
mel = torch.from_numpy(synth.synth_inputs(B, 150, seed=5)["mel"]).to(dev)
mlen = torch.full((B,), 150, dtype=torch.int64, device=dev)
mel2, mlen2 = torch.cat([mel, mel]), torch.cat([mlen, mlen])
dur = torch.full((B, N), 10.0, device=dev)
noise = torch.from_numpy(r.standard_normal((B, 192, T2)).astype(np.float32)).to(dev)


def front():
    return models.ttv.inf_extract_tc_latent(ids, tlen, mel, mlen, tone, lang, dur=dur)


x_frame, g, x_lengths, x_mask = front()


def back(codes):
    w2v, pitch = models.ttv.inf_plm_gen(x_frame, g, codes, x_lengths, x_mask)
    pitch = IP.zero_below(pitch, float(np.log(55.0)))
    frames = torch.ceil(x_lengths).to(torch.int64)
    audio = models.voc.voice_conversion_noise_control(w2v, frames, mel2, mlen2, pitch.unsqueeze(1), noise_scale=0.333,
                                                      denoise_ratio=0.0, noise=noise)
    return IP.peak_int16(audio, frames * 320)


codes0 = models.plm.infer(x_frame)
wav0 = back(codes0)
torch.cuda.synchronize()
g_plm = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_plm):
    codes_g = models.plm.infer(x_frame)
g_back = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_back):
    wav_g = back(codes_g)
g_plm.replay()
g_back.replay()
torch.cuda.synchronize()
assert torch.equal(codes_g, codes0) and torch.equal(wav_g, wav0)

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(lo, hi):
    """a stream whose queue may use the CUs [lo, hi) of the mask's bit order only"""
    words = (NCU + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


out = {"cus_total": NCU}
# ---- the sequential flow (bench_extra.tts_b16): front-end eager, PLM graph, host wait, back-half graph
def sequential():
    front()
    g_plm.replay()
    torch.cuda.current_stream().synchronize()
    g_back.replay()


out["sequential_ms_per_batch"] = timed(sequential, a.iters)
out["plm_graph_alone_ms"] = timed(g_plm.replay)
out["back_graph_alone_ms"] = timed(g_back.replay)
out["front_eager_ms"] = timed(front)
print(json.dumps({k: round(v, 2) if isinstance(v, float) else v for k, v in out.items()}), flush=True)

# ---- 1. does a graph keep the mask?  the MFMA-bound back half on N CUs
out["by_cus"] = {}
for n_cu in a.cus:
    sp, sv = masked_stream(0, n_cu), masked_stream(n_cu, NCU)
    res = {}

    def on(st, fn):
        def run():
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                fn()
            torch.cuda.current_stream().wait_stream(st)
        return run

    res["back_graph_on_masked_ms"] = timed(on(sp, g_back.replay), 2)        # ~ x NCU / n_cu if the graph keeps the mask
    res["back_graph_on_complement_ms"] = timed(on(sv, g_back.replay))
    res["plm_graph_on_masked_ms"] = timed(on(sp, g_plm.replay))
    res["plm_eager_on_masked_ms"] = timed(on(sp, lambda: models.plm.infer(x_frame)), 2)

    # ---- 3. the pipeline.  Buffers are the static graph inputs (every batch is the same synthetic batch), so the data
    # dependencies between iterations hold trivially; what is measured is the concurrency.
    def pipeline_iter():
        main = torch.cuda.current_stream()
        sp.wait_stream(main)
        sv.wait_stream(main)
        with torch.cuda.stream(sp):
            g_plm.replay()                      # batch i + 1
        with torch.cuda.stream(sv):
            front()                             # batch i + 2 (eager: it holds the reference's host read-back of T)
            g_back.replay()                     # batch i
        main.wait_stream(sp)
        main.wait_stream(sv)
        torch.cuda.synchronize()

    res["pipeline_ms_per_batch"] = timed(pipeline_iter, a.iters)

    def pipeline_graph_first():
        main = torch.cuda.current_stream()
        sp.wait_stream(main)
        sv.wait_stream(main)
        with torch.cuda.stream(sv):
            front()
            g_back.replay()
        with torch.cuda.stream(sp):
            g_plm.replay()
        main.wait_stream(sp)
        main.wait_stream(sv)
        torch.cuda.synchronize()

    res["pipeline_back_submitted_first_ms_per_batch"] = timed(pipeline_graph_first, a.iters)

    def pipeline_plm_eager():
        main = torch.cuda.current_stream()
        sp.wait_stream(main)
        sv.wait_stream(main)
        with torch.cuda.stream(sv):
            front()
            g_back.replay()
        with torch.cuda.stream(sp):
            models.plm.infer(x_frame)
        main.wait_stream(sp)
        main.wait_stream(sv)
        torch.cuda.synchronize()

    res["pipeline_plm_eager_ms_per_batch"] = timed(pipeline_plm_eager, 3)
    g_plm.replay()
    g_back.replay()
    torch.cuda.synchronize()
    assert torch.equal(codes_g, codes0) and torch.equal(wav_g, wav0)
    out["by_cus"][str(n_cu)] = res
    print(n_cu, json.dumps({k: round(v, 2) for k, v in res.items()}), flush=True)

# unmasked two-stream pipeline for reference
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def pipeline_unmasked():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    s2.wait_stream(main)
    with torch.cuda.stream(s1):
        g_plm.replay()
    with torch.cuda.stream(s2):
        front()
        g_back.replay()
    main.wait_stream(s1)
    main.wait_stream(s2)
    torch.cuda.synchronize()


out["pipeline_unmasked_streams_ms_per_batch"] = timed(pipeline_unmasked, a.iters)
print("unmasked", round(out["pipeline_unmasked_streams_ms_per_batch"], 2))
if a.json:
    with open(a.json, "w") as fh:
        json.dump(out, fh, indent=1)
