#!/usr/bin/env python3
"""BASELINE.json configs[2] (full text -> wav, batch 16) and configs[3] (vocoder + SpeechSR48, batch 32)
as functions: bench.py prints them as `extra_configs` beside the headline; tools/tts_bench.py and
tools/sr_bench.py are the stand-alone command lines.

Each function returns a dict {value, unit, ms_per_step, rtf, stage_ms, roofline, config}.  `roofline` is for the
kernel that takes most of the time among the launches behind the hsp_conv1d entry points (conv1d_mfma_kernel
or the token GEMM): algorithmic FLOP and bytes of its launches / their summed duration, from one extra eager
pass with a HIP-event pair per launch on the launch stream."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
TRAFFIC_EXTRA = "r06_traffic_extra.json"   # tools/pmc_traffic_extra.sh + tools/pmc_summarize_extra.py

VOC_CFG = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3,
               p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
               upsample_rates=[4, 5, 4, 2, 2], upsample_initial_channel=1024, upsample_kernel_sizes=[8, 11, 8, 4, 4],
               gin_channels=256)
TTV_CFG = dict(inter_channels=256, hidden_channels=256, filter_channels=1024, n_heads=4, n_layers=6, kernel_size=3,
               p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
               use_spectral_norm=False)


def event_median_ms(fn, steps):
    """median over `steps` calls of fn(), each bracketed by a HIP-event pair on the current stream"""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))


def conv_entry_profile(fn):
    """Run fn() once eagerly on ONE stream with an event pair around every launch behind the conv entry points;
    returns {kernel name: (launches, flop, bytes, ms)}."""
    from megatts2_hierspeechpp_amd import _lib as L
    from megatts2_hierspeechpp_amd import hierspeechpp_speechsynthesizer as hss
    from megatts2_hierspeechpp_amd import hip_layers
    rec = []

    def hook(kind, fl, nb, e0, e1, la):
        if kind == "hsp_conv1d_mfma_f32":
            plan = (C.c_int32 * 4)()
            L.check(L.lib().hsp_conv1d_mfma_plan(C.byref(la), C.byref(plan)), "hsp_conv1d_mfma_plan")
            kind = "conv1d_mfma_kernel" if plan[2] > 0 else ("rgemm_kernel" if plan[2] == -1 else "bgemm_kernel")
        if kind == "hsp_cprod3_f32":
            kind = "cprod3_kernel"
        if kind != "hsp_fftconv":          # the whole-conv record of a frequency-domain conv spans its three launches
            rec.append((kind, fl, nb, e0, e1))

    saved = hss.SERIAL_STREAMS
    hss.SERIAL_STREAMS = True          # the product's own launches, one after the other
    hip_layers.LAUNCH_HOOK = hook
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        hip_layers.LAUNCH_HOOK = None
        hss.SERIAL_STREAMS = saved
    agg = {}
    for kind, fl, nb, e0, e1 in rec:
        n, f, b, m = agg.get(kind, (0, 0, 0, 0.0))
        agg[kind] = (n + 1, f + fl, b + nb, m + e0.elapsed_time(e1))
    return agg


def _pmc_traffic(workload, kind, launches):
    """HBM bytes per launch of `kind` from the committed PMC profile of this workload's dominant stage
    (profiles/<TRAFFIC_EXTRA>, tools/pmc_traffic_extra.sh), or (None, reason)."""
    import json
    path = os.path.join(ROOT, "profiles", TRAFFIC_EXTRA)
    if workload is None or not os.path.exists(path):
        return None, f"no profiles/{TRAFFIC_EXTRA}"
    from megatts2_hierspeechpp_amd.build import source_id
    tj = json.load(open(path))
    if tj.get("kernel_source_sha16") != source_id():
        return None, f"profile taken on kernel sources {tj.get('kernel_source_sha16')}, this run uses {source_id()}"
    k = tj.get(workload, {}).get(kind)
    if not k:
        return None, f"no {kind} in the profile"
    if k["launches"] != launches:
        return None, f"profile has {k['launches']} launches of {kind}, this run {launches}"
    f16 = 1.94     # FETCH_SIZE under-reports 16-B-per-lane streams (r03_traffic.json calibrates 1.94 on the activation kernel)
    fac = 1.0 / 0.90 if kind == "rgemm_kernel" else f16
    total = 1024.0 * (fac * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"])
    return total / launches, (f"PMC FETCH_SIZE x {fac:.2f} + WRITE_SIZE per launch; raw FETCH {k['FETCH_SIZE_KB'] * 1024 / launches:.0f} B, "
                              f"WRITE {k['WRITE_SIZE_KB'] * 1024 / launches:.0f} B")


def dominant_roofline(agg, workload=None):
    if not agg:
        return None
    kind, (n, fl, nb, ms) = max(agg.items(), key=lambda kv: kv[1][3])
    traffic, tnote = _pmc_traffic(workload, kind, n)
    tf = fl / (ms * 1e-3) / 1e12
    gbs = nb / (ms * 1e-3) / 1e9
    mfma_frac, hbm_frac = tf / FP32_MFMA_PEAK_TFLOPS, gbs / HBM_PEAK_GBS
    bound = "mfma" if mfma_frac >= hbm_frac else "hbm"
    return {"kernel": kind, "bound": bound,
            "achieved": tf if bound == "mfma" else gbs, "peak": FP32_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": max(mfma_frac, hbm_frac),
            "mfma_frac": mfma_frac, "hbm_frac": hbm_frac, "launches_per_step": n, "kernel_ms_per_step": ms,
            "algorithmic_gflop_per_step": fl / 1e9, "algorithmic_mb_per_step": nb / 1e6,
            "algorithmic_bytes_per_launch": nb / n, "traffic": traffic, "traffic_note": tnote,
            "timing": "one extra eager step on one stream, event pair per launch"}


# ----------------------------------------------------------------------------- configs[2]
def tts_b16(dev, steps=3, warmup=1, batch=16, phones=40, use_graph=True, models=None):
    """16 utterances x 40 phones x 10 frames (durations pinned) -> 200 PLM steps, 4 s each."""
    from megatts2_hierspeechpp_amd import inference_plm as IP, synth
    B, N = batch, phones
    if models is None:
        models = IP.TtsModels(VOC_CFG, TTV_CFG)
        models.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0))
                                for k, v in models.state_dict().items()})
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
    T2 = N * 10 // 2
    noise = torch.from_numpy(r.standard_normal((B, 192, T2)).astype(np.float32)).to(dev)
    plm_graph = {}

    def plm_infer(x_frame, eager=False):
        if eager or not use_graph:
            return models.plm.infer(x_frame)
        if "g" not in plm_graph:
            plm_graph["x"] = x_frame.clone()
            models.plm.infer(plm_graph["x"])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                plm_graph["codes"] = models.plm.infer(plm_graph["x"])
            plm_graph["g"] = g
        plm_graph["x"].copy_(x_frame)
        plm_graph["g"].replay()
        return plm_graph["codes"]

    def back_half(x_frame, g, codes, x_lengths, x_mask, mark):
        """everything behind the PLM loop: w2v / pitch decoder, pitch clipping, vocoder, int16"""
        w2v, pitch = models.ttv.inf_plm_gen(x_frame, g, codes, x_lengths, x_mask)
        pitch = IP.zero_below(pitch, float(np.log(55.0)))
        mark()
        frames = torch.ceil(x_lengths).to(torch.int64)
        audio = models.voc.voice_conversion_noise_control(w2v, frames, mel2, mlen2, pitch.unsqueeze(1),
                                                          noise_scale=0.333, denoise_ratio=0.0, noise=noise)
        mark()
        wav = IP.peak_int16(audio, frames * 320)
        mark()
        return wav

    back_graph = {}

    def back_half_graph(x_frame, g, codes, x_lengths, x_mask):
        """the same launches replayed from a hipGraph (shapes are fixed once the front-end has read back T; the
        front-end itself holds the reference's host synchronisation and stays eager)"""
        if "g" not in back_graph:
            st = back_graph["in"] = [t.clone() for t in (x_frame, g, codes, x_lengths, x_mask)]
            back_half(*st, lambda: None)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                back_graph["wav"] = back_half(*st, lambda: None)
            back_graph["g"] = gr
        for dst, src in zip(back_graph["in"], (x_frame, g, codes, x_lengths, x_mask)):
            dst.copy_(src)
        back_graph["g"].replay()
        return back_graph["wav"]

    def step(ev=None, eager=False):
        mark = (lambda: ev.append(torch.cuda.Event(enable_timing=True)) or ev[-1].record()) if ev is not None \
            else (lambda: None)
        mark()
        x_frame, g, x_lengths, x_mask = models.ttv.inf_extract_tc_latent(ids, tlen, mel, mlen, tone, lang, dur=dur)
        mark()
        codes = plm_infer(x_frame, eager)
        mark()
        if ev is not None and use_graph:
            torch.cuda.current_stream().synchronize()     # as in the timed flow: nothing is submitted behind a running PLM graph
        if ev is None and not eager and use_graph:
            # Measured (tools/tts_graph_probe.py): anything the host submits BEHIND the running 4 600-node PLM graph --
            # eager launches or a second graph -- slows that graph down (PLM + back half 163 ms; one combined graph
            # 149.5); with the host idle until the PLM graph has drained, then ONE graph launch for the back half: 137.
            torch.cuda.current_stream().synchronize()
            return back_half_graph(x_frame, g, codes, x_lengths, x_mask)
        return back_half(x_frame, g, codes, x_lengths, x_mask, mark)

    for _ in range(max(warmup, 1)):
        wav = step()
    torch.cuda.synchronize()
    assert wav.shape == (B, 320 * T2), wav.shape
    t0 = time.perf_counter()
    for _ in range(steps):
        wav = step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    ev = []
    step(ev)
    torch.cuda.synchronize()
    names = ["front_end(A16-A17)", "plm_loop(A18)", "w2v+pitch(A17)", "vocoder(A1-A14)", "int16_post(A19)"]
    stages = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
    roof = dominant_roofline(conv_entry_profile(lambda: models.plm.infer(plm_graph["x"]) if "x" in plm_graph
                                                else step(None, eager=True)), "tts")
    if roof:
        roof["scope"] = "the PLM greedy loop (the dominant stage of this config), all launches behind the conv entry point"
    return {"metric": "16kHz audio samples/sec, full inference_plm.py text->wav, batch=16 (BASELINE.json configs[2])",
            "value": B * 320 * T2 / el, "unit": "samples/s", "ms_per_step": el * 1e3,
            "rtf": el / (B * 320 * T2 / 16000.0), "n_gpus": 1, "dtype": "f32", "data": "synthetic", "steps": steps,
            "config": {"workload": f"tts: {B} utterances x {N} phones x 10 frames -> {320 * T2 / 16000:g} s each, "
                                   "prompt mel 150 frames", "plm_steps": T2,
                       "launch_mode": ("front-end eager (it holds the reference's host read-back of T); the PLM loop and "
                                       "everything behind it are two hipGraph replays, the host waiting for the first "
                                       "before it launches the second") if use_graph else "eager"},
            "stage_ms": stages,
            "stage_ms_note": "one extra step with EAGER launches behind the PLM graph so that events can sit between the stages "
                             "(the timed flow replays that half from one hipGraph, ~6 ms faster than the eager sum)",
            "roofline": roof}


# ----------------------------------------------------------------------------- SURVEY 8(f) N2 / N4 on the bench line
def _speechlike(n, seed):
    """a deterministic speech-like waveform (harmonics of a gliding pitch + noise, |x| < 1): the prompt / source audio"""
    r = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    f0 = 120.0 + 40.0 * np.sin(2 * np.pi * 0.7 * t)
    ph = 2 * np.pi * np.cumsum(f0) / 16000.0
    x = sum(np.sin(k * ph) / k for k in range(1, 6)) * (0.5 + 0.5 * np.sin(2 * np.pi * 2.1 * t) ** 2)
    x = 0.25 * x + 0.02 * r.standard_normal(n)
    return (x / max(1.0, np.abs(x).max() / 0.95)).astype(np.float32)[None]


def vc_b1_4s(dev, steps=10, seconds=4.0, prompt_seconds=3.0):
    """The reference's second way into the vocoder (inference_vc.py:70-145): ONE source utterance of `seconds` of 16 kHz audio
    -> reflect pad -> wav2vec2 (MMS-300M topology) hidden state 7 -> F0 conversion against the prompt's track ->
    prompt mels -> voice_conversion_noise_control -> int16.  The YAAPT F0 tracks are inputs (CPU numpy code of a third-party
    package in the reference; absent here), drawn like synth_inputs' f0.  hipGraph replay of the whole tensor core; stage
    split from an eager pass with events; `roofline` = the dominant kernel of the wav2vec2 producer behind the conv entry points."""
    from megatts2_hierspeechpp_amd import functional as Fh, inference_vc as IV, synth
    from megatts2_hierspeechpp_amd.Mels_preprocess import MelSpectrogramFixed
    from megatts2_hierspeechpp_amd.inference_plm import peak_int16
    models = IV.VcModels(VOC_CFG)
    models.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in models.state_dict().items()})
    models.finalize(dev)
    mel_fn = MelSpectrogramFixed(sample_rate=16000, n_fft=1280, win_length=1280, hop_length=320, f_min=0, f_max=8000,
                                 n_mels=80, window_fn=torch.hann_window).finalize(dev)
    n_src = int(seconds * 16000) - 640                       # pad_source always adds: 4 s after padding
    src = IV.pad_source(torch.from_numpy(_speechlike(n_src, 11)).to(dev))
    trg = torch.from_numpy(_speechlike(int(prompt_seconds * 16000), 12)).to(dev)
    T = src.shape[-1] // 320
    r = np.random.default_rng(13)
    mk_f0 = lambda n: torch.from_numpy(np.where(r.random((1, n)) < 0.3, 0, r.uniform(90, 300, (1, n))).astype(np.float32)).to(dev)
    f0s, f0t = mk_f0(4 * T), mk_f0(trg.shape[-1] // 80)
    noise = torch.from_numpy(r.standard_normal((1, 192, T)).astype(np.float32)).to(dev)

    def stages(mark):
        mark()
        x_w2v = models.w2v(Fh.reflect_pad(src, 40))
        mark()
        lf0 = Fh.f0_convert(f0s, f0t)
        both = torch.cat([trg, trg], 0)
        trg_mel = mel_fn(both)
        mark()
        xl = torch.full((1,), T, dtype=torch.int64, device=dev)
        tl = torch.full((2,), trg_mel.shape[2], dtype=torch.int64, device=dev)
        audio = models.voc.voice_conversion_noise_control(x_w2v, xl, trg_mel, tl, lf0.reshape(1, -1)[:, :4 * T],
                                                          noise_scale=0.333, denoise_ratio=0.0, noise=noise)
        mark()
        wav = peak_int16(audio.reshape(1, -1), torch.full((1,), audio.shape[-1], dtype=torch.int64, device=dev))
        mark()
        return wav

    wav = stages(lambda: None)
    torch.cuda.synchronize()
    assert wav.shape == (1, 320 * T) and wav.dtype == torch.int16
    # the harness entry point itself gives the same samples (it builds its length tensors on the host)
    ref = IV.vc(models, mel_fn, src, f0s, trg, f0t, noise_scale_vc=0.333, denoise_ratio=0.0, noise=noise)
    assert torch.equal(ref, wav.reshape(-1))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        stages(lambda: None)
    g.replay()
    torch.cuda.synchronize()
    ms = event_median_ms(g.replay, steps)
    ev = []
    stages(lambda: ev.append(torch.cuda.Event(enable_timing=True)) or ev[-1].record())
    torch.cuda.synchronize()
    names = ["wav2vec2_hidden7(N2)", "f0_convert+prompt_mels(N1,N2)", "vocoder(A1-A14)", "int16_post(A19)"]
    stage_ms = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
    gw = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gw):
        models.w2v(Fh.reflect_pad(src, 40))
    gw.replay()
    torch.cuda.synchronize()
    w2v_ms = event_median_ms(gw.replay, steps)
    roof = dominant_roofline(conv_entry_profile(lambda: models.w2v(Fh.reflect_pad(src, 40))), "vc_w2v")
    if roof:
        roof["scope"] = ("the wav2vec2 producer only (7 strided feature convs at 512 channels + positional conv + 7 transformer "
                         "layers of 1024 / 4096): the dominant kernel among the launches behind the conv entry points")
    return {"metric": "latency of the inference_vc.py tensor core, 1 utterance x 4 s (SURVEY 8(f) N2)",
            "value": 320 * T / (ms * 1e-3), "unit": "samples/s", "ms_per_step": ms, "rtf": ms * 1e-3 / (T / 50.0),
            "n_gpus": 1, "dtype": "f32", "data": "synthetic", "steps": steps,
            "config": {"workload": f"vc: 1 x {T / 50:g} s source (reflect pad 40 -> wav2vec2 hidden state 7, {T} frames), "
                                   f"{prompt_seconds:g}-s prompt, F0 tracks given",
                       "launch_mode": "hipGraph replay of the whole tensor core, median of HIP-event pairs"},
            "stage_ms": stage_ms, "stage_ms_note": "one eager pass with events between the stages (host submission included)",
            "w2v_producer_graph_ms": w2v_ms, "roofline": roof}


def tts_prompt_denoise(dev, steps=3, models=None, prompt_seconds=3.0, phones=40):
    """inference_plm.py's default call (denoise_ratio = 0.8, :308-309): prompt WAVEFORM -> MP-SENet denoiser (denoiser/infer.py:4-33)
    -> two prompt mels -> text -> w2v / f0 -> vocoder with the mixed style vector -> int16; ONE utterance of 40 phones x 10
    frames = 4 s, as the reference runs it.  Eager launches (the flow holds the reference's host read-backs); the denoiser's
    share is timed alone (eager, event pairs) and the `roofline` is its dominant kernel behind the conv entry points."""
    import types
    from megatts2_hierspeechpp_amd import inference_plm as IP, synth
    from megatts2_hierspeechpp_amd.Mels_preprocess import MelSpectrogramFixed
    from megatts2_hierspeechpp_amd.denoiser.generator import MPNet
    from megatts2_hierspeechpp_amd.denoiser.infer import denoise
    hd = types.SimpleNamespace(dense_channel=64, compress_factor=0.3, num_tsconformers=4, beta=2.0, sampling_rate=16000,
                               n_fft=400, hop_size=100, win_size=400)   # denoiser/config.json of the reference
    if models is None:
        models = IP.TtsModels(VOC_CFG, TTV_CFG)
        models.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0)) for k, v in models.state_dict().items()})
        models.finalize(dev)
    den = MPNet(hd)
    den.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 7)) for k, v in den.state_dict().items()})
    den.finalize(dev)
    mel_fn = MelSpectrogramFixed(sample_rate=16000, n_fft=1280, win_length=1280, hop_length=320, f_min=0, f_max=8000,
                                 n_mels=80, window_fn=torch.hann_window).finalize(dev)
    r = np.random.default_rng(3)
    N = phones
    ids = torch.from_numpy(r.integers(12, 113, (1, N))).to(dev)
    tone = torch.from_numpy(r.integers(0, 11, (1, N))).to(dev)
    lang = torch.where(ids < 74, 1, 2).to(dev)
    prompt = torch.from_numpy(_speechlike(int(prompt_seconds * 16000), 21)).to(dev)
    dur = torch.full((1, N), 10.0, device=dev)
    T2 = N * 10 // 2
    noise = torch.from_numpy(r.standard_normal((1, 192, T2)).astype(np.float32)).to(dev)
    run = lambda ratio: IP.tts_from_prompt(models, mel_fn, ids, tone, lang, prompt, dur=dur, noise=noise, denoise_ratio=ratio,
                                           denoiser=den if ratio else None, hps_denoiser=hd)
    wav = run(0.8)
    torch.cuda.synchronize()
    assert wav.shape == (320 * T2,) and wav.dtype == torch.int16

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    ms = timed(lambda: run(0.8))
    ms_plain = timed(lambda: run(0.0))
    padded = torch.zeros((prompt.shape[-1] // 1600 + 1) * 1600, device=dev)
    padded[:prompt.shape[-1]].copy_(prompt[0])
    denoise(padded, den, hd)
    torch.cuda.synchronize()
    # eager: denoise() reads the prompt's energy back for its norm factor (denoiser/infer.py:5: a host value in the reference too)
    den_ms = event_median_ms(lambda: denoise(padded, den, hd), 10)
    roof = dominant_roofline(conv_entry_profile(lambda: denoise(padded, den, hd)), "denoiser")
    if roof:
        roof["scope"] = ("the prompt denoiser only (MP-SENet: dense encoder, 4 two-stage conformers over 161 time x 201 frequency "
                         "positions per second, two decoders): the dominant kernel among the launches behind the conv entry points")
    return {"metric": "latency of inference_plm.py text->wav from the prompt waveform with denoise_ratio = 0.8, 1 utterance x 4 s "
                      "(SURVEY 8(f) N4 + N1)",
            "value": 320 * T2 / (ms * 1e-3), "unit": "samples/s", "ms_per_step": ms, "rtf": ms * 1e-3 / (T2 / 50.0),
            "n_gpus": 1, "dtype": "f32", "data": "synthetic", "steps": steps,
            "config": {"workload": f"tts_from_prompt: {prompt_seconds:g}-s prompt waveform -> denoiser -> prompt mels -> {N} phones x 10 "
                                   f"frames -> {T2 / 50:g} s", "launch_mode": "eager (the reference's host read-backs stay)"},
            "ms_per_step_without_denoiser": ms_plain, "denoiser_ms": den_ms,
            "denoiser_share": den_ms / ms, "roofline": roof}


# ----------------------------------------------------------------------------- configs[3]
def vocoder_b1(dev, seconds=1.0, steps=20, net=None):
    """Latency of ONE utterance through vocoder-only infer() -- batch 1 is the only way the reference itself runs
    (inference_plm.py:277-287 loops over sentences one at a time) -- hipGraph replay, median of HIP-event pairs."""
    out = vocoder_b1_1s(dev, steps=steps, net=net, frames=int(round(seconds * 50)))
    out["metric"] = f"latency of vocoder-only infer(), 1 utterance x {seconds:g} s"
    out["config"]["workload"] = f"vocoder infer() 1 x {seconds:g} s ({int(round(seconds * 50))} frames)"
    return out


def tts_b1(dev, steps=3, models=None):
    """Latency of ONE sentence through the whole text -> wav chain (40 phones x 10 frames = 4 s), the reference's own
    usage (inference_plm.py:277-287): tts_b16's flow at batch 1."""
    out = tts_b16(dev, steps=steps, warmup=1, batch=1, models=models)
    out["metric"] = "latency of full inference_plm.py text->wav, 1 utterance x 4 s"
    return out


def tts_2x16(dev, steps=3, models=None):
    """Two 16-utterance TTS batches in flight, as ONE 32-row pass.  The greedy PLM loop is bound by
    launch floors -- its per-step time does not depend on the step index or (much) on the row count
    (profiles/r05_plm_step_curve.txt: 418 us per step at 16 rows, 391 at 32) -- so a second batch rides along for 27 ms instead
    of 70, and the back half runs at the vocoder's 32-utterance rate.  What does NOT work is hiding the loop of batch i + 1
    under the vocoder of batch i on a CU-masked stream (hipExtStreamCreateWithCUMask; profiles/r05_plm_overlap_probe.json):
    the loop's launches are hundreds of workgroups wide, on 64 CUs it takes 243 ms instead of 70.  `ms_per_batch_of_16` is
    the throughput figure; the latency of a single batch stays tts_b16's."""
    out = tts_b16(dev, steps=steps, warmup=1, batch=32, models=models)
    out["metric"] = "16kHz audio samples/sec, full inference_plm.py text->wav, two batches of 16 in flight as one 32-row pass"
    out["ms_per_batch_of_16"] = out["ms_per_step"] / 2.0
    return out


def vocoder_b1_1s(dev, steps=20, net=None, frames=50):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case: vocoder-only infer(), 1 utterance x 1 s) on
    the GPU: the latency of one small request, hipGraph replay.  bench.py's cpu_baseline times the same case on the
    host (config0_1x1s)."""
    from megatts2_hierspeechpp_amd import synth
    from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import SynthesizerTrn
    if net is None:
        net = SynthesizerTrn(641, 192, **VOC_CFG)
        net.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0))
                             for k, v in net.state_dict().items()})
        net.finalize(dev)
    B, T = 1, frames
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_inputs(B, T, seed=2).items()}

    def step():
        return net.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])[0]

    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = step()
    g.replay()
    torch.cuda.synchronize()
    ms = event_median_ms(g.replay, steps)
    assert o.shape == (B, 1, 320 * T) and bool(torch.isfinite(o).all())
    return {"metric": "latency of vocoder-only infer(), 1 utterance x 1 s (BASELINE.json configs[0])",
            "value": B * 320 * T / (ms * 1e-3), "unit": "samples/s", "ms_per_step": ms, "rtf": ms * 1e-3 / (T / 50.0),
            "n_gpus": 1, "dtype": "f32", "data": "synthetic", "steps": steps,
            "config": {"workload": "vocoder infer() 1 x 1 s (50 frames)", "launch_mode": "hipGraph replay, median of HIP-event pairs"}}


def sr48_b32(dev, steps=5, batch=32, net=None):
    """vocoder (32 x 4 s) -> SpeechSR48: 48 kHz output samples / s of the two-stage pipeline, hipGraph replay."""
    from megatts2_hierspeechpp_amd import synth
    from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import SynthesizerTrn
    from megatts2_hierspeechpp_amd.hip_layers import finalize
    from megatts2_hierspeechpp_amd.speechsr48k.speechsr import SynthesizerTrn as SpeechSR
    sr = SpeechSR(128, 30, "0", [3, 7, 11], [[1, 3, 5]] * 3, [3], 32, [3])
    if net is None:
        net = SynthesizerTrn(641, 192, **VOC_CFG)
        net.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0))
                             for k, v in net.state_dict().items()})
        net.finalize(dev)
    sr.load_state_dict({k: torch.from_numpy(synth.synth_tensor("sr." + k, tuple(v.shape), 0))
                        for k, v in sr.state_dict().items()})
    finalize(sr, dev)
    B, T = batch, 200
    inp = synth.synth_inputs(B, T, seed=1)
    d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}

    def step():
        o, _ = net.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])
        return o, sr(o)

    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o16, o48 = step()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    # stage split: each stage captured into its OWN hipGraph and replayed (device time; round 3 timed one eager pass
    # with events, which measured host submission and first-size allocations: 157 + 296 ms inside a 101-ms step)
    gv, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gv):
        o, _ = net.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])
    with torch.cuda.graph(gs):
        sr(o)
    for gg in (gv, gs):
        gg.replay()
    torch.cuda.synchronize()
    ms_v, ms_s = event_median_ms(gv.replay, max(steps, 3)), event_median_ms(gs.replay, max(steps, 3))
    assert o48.shape == (B, 1, 3 * 320 * T) and bool(torch.isfinite(o48).all())
    roof = dominant_roofline(conv_entry_profile(lambda: sr(o)), "sr48")
    if roof:
        roof["scope"] = "SpeechSR48 stage only (the vocoder stage is the headline's roofline)"
    return {"metric": "48 kHz samples/s, vocoder + SpeechSR48, batch=32 (BASELINE.json configs[3])",
            "value": B * 3 * 320 * T / el, "unit": "samples/s", "ms_per_step": el * 1e3,
            "rtf": el / (B * 320 * T / 16000.0), "n_gpus": 1, "dtype": "f32", "data": "synthetic", "steps": steps,
            "config": {"workload": f"vocoder infer() {B} x 4 s -> SpeechSR48 (x3 linear interp + AMP block, C=32)",
                       "launch_mode": "hipGraph replay of both stages"},
            "stage_ms": {"vocoder": ms_v, "speechsr48": ms_s},
            "stage_ms_note": "each stage replayed from its own hipGraph, median of HIP-event pairs",
            "roofline": roof}
