"""Shared test plumbing: golden-fixture loading and the fixture-kind -> module mapping.

Golden fixtures (tests/golden/*.npz) hold outputs of the REFERENCE run on CPU by
tools/make_golden.py; weights are regenerated from the synthetic recipe."""
from __future__ import annotations

import glob
import json
import os

import numpy as np
import torch

from megatts2_hierspeechpp_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# fp32 parity bar of BASELINE.json:north_star: 1e-4 on outputs of O(1) magnitude;
# scaled by the reference's peak magnitude for outputs that are not audio
ATOL = 1e-4


# ttv_v1/config.json "model" section of the reference (hyper-parameters, not code)
TTV_MODEL = dict(inter_channels=256, hidden_channels=256, filter_channels=1024, n_heads=4, n_layers=6, kernel_size=3,
                 p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11],
                 resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], use_spectral_norm=False)


# denoiser/config.json of the reference (hyper-parameters, not code)
import types  # noqa: E402
DENOISER_H = types.SimpleNamespace(dense_channel=64, compress_factor=0.3, num_tsconformers=4, beta=2.0,
                                   sampling_rate=16000, n_fft=400, hop_size=100, win_size=400)


def fixture_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    arrays = {k: z[k] for k in z.files if k != "meta"}
    return meta, arrays


def synth_sd(meta, as_torch=True):
    pre = meta["prefix"] + "." if meta["prefix"] else ""
    sd = {k: synth.synth_tensor(pre + k, tuple(s), meta["seed"]) for k, s in meta["shapes"]}
    return {k: torch.from_numpy(v) for k, v in sd.items()} if as_torch else sd


def oracle_sd(meta):
    """state dict keyed the way oracle functions address it (prefix.key)."""
    pre = meta["prefix"] + "." if meta["prefix"] else ""
    return {pre + k: v for k, v in synth_sd(meta).items()}


def tol_for(ref: np.ndarray) -> float:
    return ATOL * max(1.0, float(np.abs(ref).max()))


def outputs(arrays):
    return [arrays[k] for k in sorted(k for k in arrays if k.startswith("out"))]


# ------------------------------------------------------------------ oracle side
def run_oracle(meta, arrays):
    from oracle import hsp_oracle as O
    sd = oracle_sd(meta)
    t = lambda k: torch.from_numpy(arrays[k])
    kind, name = meta["kind"], meta["prefix"]
    cfg = O.default_config()
    mask = None
    if "lengths" in arrays:
        T = {"dit_block": lambda: arrays["x"].shape[1], "style_encoder": lambda: arrays["mel"].shape[2],
             "posterior_sf": lambda: arrays["w2v"].shape[2], "infer": lambda: arrays["mel"].shape[2],
             "plm": lambda: arrays["tc"].shape[2], "ttv_front": lambda: arrays["ids"].shape[1],
             "ttv_gen": lambda: arrays["x_frame"].shape[2]}.get(
            kind, lambda: arrays["x"].shape[2])()
        mask = O.sequence_mask(t("lengths"), T).unsqueeze(1).float()
    if kind == "act1d":
        return [O.act1d(sd, name, t("x"))]
    if kind == "amp_block":
        return [O.amp_block(sd, name, t("x"), meta["k"])]
    if kind == "convtr":
        return [O.conv_transpose1d(sd, name, t("x"), meta["u"], (meta["k"] - meta["u"]) // 2)]
    if kind == "dblock":
        return [O.dblock(sd, name, t("x"))]
    if kind == "rel_mha":
        am = mask.unsqueeze(2) * mask.unsqueeze(-1)
        return [O.mha_relpos(sd, name, t("x"), t("x"), am, meta["heads"], meta["window"])]
    if kind == "vits_encoder":
        return [O.vits_encoder(sd, name, t("x"), mask, meta["heads"], meta["layers"], meta["k"], meta["window"])]
    if kind == "wn":
        return [O.wavenet(sd, name, t("x"), mask, t("g"), meta["hidden"], meta["k"], meta["n_layers"])]
    if kind == "dit_block":
        return [O.dit_conv_block(sd, name, t("x"), t("c"), mask.transpose(1, 2))]
    if kind == "coupling":
        return [O.coupling_reverse(sd, name, t("x"), mask, t("c"))]
    if kind == "flow":
        return [O.flow_reverse(sd, name, t("x"), mask, t("g"))]
    if kind == "style_encoder":
        return [O.style_encoder(sd, name, t("mel"), mask)]
    if kind == "posterior_sf":
        return list(O.posterior_sf_encoder(sd, name, t("w2v"), t("f0"), mask, t("g"), t("noise")))
    if kind == "source_network":
        return list(O.source_network(sd, name, t("z"), t("g")))
    if kind == "generator":
        return [O.generator(sd, name, t("z"), t("e"), t("g"), cfg)]
    if kind == "infer":
        return list(O.synth_infer(sd, cfg, t("mel"), t("w2v"), t("lengths"), t("f0"), t("noise")))
    if kind == "vc":
        return [O.synth_voice_conversion_noise_control(sd, cfg, t("w2v"), t("src_length"), t("mel"), t("trg_length"),
                                                       t("f0"), meta["noise_scale"], meta["denoise_ratio"], t("noise"))]
    if kind == "tts_e2e":
        sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        n, tm = arrays["ids"].shape[1], arrays["src_mel"].shape[2]
        wav, audio = O.tts_one(sub("ttv."), {k: v for k, v in sd.items() if k.startswith("plm.")}, sub("voc."), cfg,
                               t("ids"), t("tone"), t("language"), t("mel_ttv"), t("src_mel"),
                               torch.tensor([tm, tm]), t("noise"), meta["noise_scale"], meta["denoise_ratio"])
        return [audio, torch.from_numpy(wav.astype(np.float32))]
    if kind == "ttv_front":
        per = []
        for b, (n, tm) in enumerate(zip(arrays["lengths"], arrays["mel_lengths"])):
            xf, g, fl, _ = O.ttv_extract_tc_latent_one(sd, t("ids")[b:b + 1, :n], t("mel")[b:b + 1, :, :tm],
                                                        t("tone")[b:b + 1, :n], t("language")[b:b + 1, :n])
            per.append((xf, g, torch.tensor([fl])))
        T2 = max(p[0].shape[2] for p in per)
        return [torch.cat([torch.nn.functional.pad(p[0], (0, T2 - p[0].shape[2])) for p in per]),
                torch.cat([p[1] for p in per]), torch.cat([p[2] for p in per])]
    if kind == "ttv_gen":
        xf, fl = t("x_frame"), arrays["frame_lengths"]
        w2v, lf0 = torch.zeros(xf.shape[0], 1024, xf.shape[2]), torch.zeros(xf.shape[0], 4 * xf.shape[2])
        for b, n in enumerate(arrays["lengths"]):
            w, l0 = O.ttv_plm_gen_one(sd, xf[b:b + 1, :, :n], t("g")[b:b + 1], t("codes")[b:b + 1, :n], float(fl[b]))
            w2v[b, :, :n], lf0[b, :4 * n] = w[0], l0[0]
        return [w2v, lf0]
    if kind == "plm":
        # per utterance, as the reference loop runs (B = 1); positions past a length are -1 / 0
        tc, lens = t("tc"), arrays["lengths"]
        codes = torch.full(tc.shape[::2], -1.0)
        logits = torch.zeros(tc.shape[0], tc.shape[2], 1024)
        for b, n in enumerate(lens):
            c, lg = O.plm_infer(sd, name, tc[b:b + 1, :, :n], return_logits=True)
            codes[b, :n], logits[b, :n] = c[0].float(), lg[0]
        return [codes, logits]
    if kind == "speechsr":
        return [O.speechsr(sd, t("x"), meta["factor"], name + ".dec")]
    if kind == "speechsr_real":   # the reference's shipped checkpoint travels inside the fixture ("w:" arrays)
        real = {k[2:]: torch.from_numpy(v) for k, v in arrays.items() if k.startswith("w:")}
        return [O.speechsr(real, t("x"), meta["factor"], "dec")]
    if kind == "vc_plain":
        return [O.synth_voice_conversion(sd, cfg, t("w2v"), t("src_length"), t("mel"), t("trg_length"), t("f0"),
                                         meta["noise_scale"], t("noise"), uncond=bool(meta.get("uncond", False)))]
    if kind == "w2v":
        # synth_sd holds the renamed (weight_g / weight_v) keys the product module uses; the oracle accepts both spellings
        return [O.wav2vec2_hidden(sd, t("x"), meta["layer"])]
    if kind == "ttv_infer":
        w2v, lf0, _ = O.ttv_infer_one(sd, t("ids"), t("mel"), t("tone"), t("language"), t("dur"))
        return [w2v, lf0]
    if kind == "mp_conformer":
        return [O.mp_conformer_block(sd, name, t("x"))]
    if kind == "mp_dense_encoder":
        return [O.mp_dense_encoder(sd, name, t("x"))]
    if kind == "denoise":
        audio, amp, _ = O.denoise(sd, name, t("wav"), spectrogram=(t("amp_in"), t("pha_in")))
        return [audio, amp]
    raise KeyError(kind)


# --------------------------------------------------------------------- HIP side
def build_module(meta):
    """The product-side module matching a fixture kind (CPU construction only)."""
    from oracle.hsp_oracle import default_config
    from megatts2_hierspeechpp_amd import activations, modules
    from megatts2_hierspeechpp_amd import hierspeechpp_speechsynthesizer as H
    from megatts2_hierspeechpp_amd.alias_free_torch import Activation1d
    from megatts2_hierspeechpp_amd.hip_layers import ConvTranspose1d
    from megatts2_hierspeechpp_amd.styleencoder import StyleEncoder
    kind = meta["kind"]
    cfg = default_config()
    shapes = dict((k, tuple(s)) for k, s in meta["shapes"])
    if kind == "act1d":
        return Activation1d(activations.SnakeBeta(shapes["act.alpha"][0], alpha_logscale=True))
    if kind == "amp_block":
        return H.AMPBlock1(meta["C"], meta["k"], (1, 3, 5))
    if kind == "convtr":
        return ConvTranspose1d(meta["ci"], meta["co"], meta["k"], meta["u"], padding=(meta["k"] - meta["u"]) // 2,
                               weight_norm=True)
    if kind == "dblock":
        return H.DBlock(16, 64, 4)
    if kind == "wn":
        return modules.WN(meta["hidden"], meta["k"], 1, meta["n_layers"], gin_channels=256)
    if kind == "dit_block":
        return modules.DiTConVBlock(192, 2, mlp_ratio=4.0, kernel=5)
    if kind == "coupling":
        return modules.ResidualCouplingLayer_Transformer_simple(192, 192, 5, 1, 3, mean_only=True)
    if kind == "flow":
        return H.ResidualCouplingBlock_Transformer(192, 192, 5, 1, 3, gin_channels=256)
    if kind == "style_encoder":
        return StyleEncoder(in_dim=80, hidden_dim=256, out_dim=256)
    if kind == "posterior_sf":
        return H.PosteriorSFEncoder(1024, 192, 192, 5, 1, 16, gin_channels=256)
    if kind == "source_network":
        return H.SourceNetwork(512)
    if kind == "generator":
        return H.Generator(192, cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"], cfg["upsample_rates"],
                           cfg["upsample_initial_channel"], cfg["upsample_kernel_sizes"], gin_channels=256)
    if kind in ("infer", "vc", "vc_plain"):
        return H.SynthesizerTrn(641, 61440 // 320, cfg=bool(meta.get("cfg", False)), **cfg)
    if kind == "rel_mha":
        from megatts2_hierspeechpp_amd import attentions
        return attentions.MultiHeadAttention(256, 256, meta["heads"], window_size=meta["window"])
    if kind == "vits_encoder":
        from megatts2_hierspeechpp_amd import attentions
        return attentions.Encoder(256, 1024, meta["heads"], meta["layers"], kernel_size=meta["k"], window_size=meta["window"])
    if kind == "plm":
        from megatts2_hierspeechpp_amd.ttv_v1.t2w2v_transformer import Megatts2PLM1
        return Megatts2PLM1()
    if kind == "tts_e2e":
        from megatts2_hierspeechpp_amd.inference_plm import TtsModels
        return TtsModels(cfg, TTV_MODEL)
    if kind in ("ttv_front", "ttv_gen", "ttv_infer"):
        from megatts2_hierspeechpp_amd.ttv_v1.t2w2v_transformer import SynthesizerTrn as Text2W2V
        return Text2W2V(126, 11, 4, 641, 320, 16000, 60, **TTV_MODEL)
    if kind == "w2v":
        from megatts2_hierspeechpp_amd.extract_w2v import Wav2vec2
        return Wav2vec2(layer=meta["layer"])
    if kind in ("mp_conformer", "mp_dense_encoder", "denoise"):
        from megatts2_hierspeechpp_amd.denoiser import conformer, generator
        h = DENOISER_H
        if kind == "mp_conformer":
            return conformer.ConformerBlock(dim=64, n_head=4, ccm_kernel_size=31)
        return generator.DenseEncoder(h, in_channel=2) if kind == "mp_dense_encoder" else generator.MPNet(h)
    if kind in ("speechsr", "speechsr_real"):
        if meta["factor"] == 1.5:    # the 24 kHz model pins x1.5 whatever its config says (speechsr24k/speechsr.py:96)
            from megatts2_hierspeechpp_amd.speechsr24k.speechsr import SynthesizerTrn as SR
            return SR(128, 30, "0", [3, 7, 11], [[1, 3, 5]] * 3, [3], 32, [3])
        from megatts2_hierspeechpp_amd.speechsr48k.speechsr import SynthesizerTrn as SR
        return SR(128, 30, "0", [3, 7, 11], [[1, 3, 5]] * 3, [meta["factor"]], 32, [3])
    raise KeyError(kind)


def run_hip(meta, arrays, device, drop_in=False):
    """Run a fixture through the HIP product path; returns a list of numpy outputs.
    ``drop_in``: the reference's own call sequence (inference_plm.py:215-262) -- ``Model(...).cuda()``,
    ``load_state_dict``, ``.eval()``, inference call -- with no finalize() anywhere."""
    from megatts2_hierspeechpp_amd import functional as Fh
    from megatts2_hierspeechpp_amd.hip_layers import finalize
    mod = build_module(meta)
    if meta["kind"] == "speechsr_real":
        sd = {k[2:]: torch.from_numpy(v) for k, v in arrays.items() if k.startswith("w:")}
    else:
        sd = synth_sd(meta)
        # parameters the fixture's path never reads (sub-modules of other entry points of the same class) keep the
        # synthetic recipe too, so that the load stays strict
        pre = meta["prefix"] + "." if meta["prefix"] else ""
        for k, v in mod.state_dict().items():
            if k not in sd:
                sd[k] = torch.from_numpy(synth.synth_tensor(pre + k, tuple(v.shape), meta["seed"]))
    if drop_in:
        mod = mod.cuda(device)
        mod.load_state_dict(sd, strict=True)
        _ = mod.eval()
    else:
        mod.load_state_dict(sd, strict=True)
        finalize(mod, device)
    d = lambda k: torch.from_numpy(arrays[k]).to(device)
    kind = meta["kind"]
    mask = None
    if "lengths" in arrays:
        T = {"dit_block": lambda: arrays["x"].shape[1], "style_encoder": lambda: arrays["mel"].shape[2],
             "posterior_sf": lambda: arrays["w2v"].shape[2], "infer": lambda: arrays["mel"].shape[2],
             "plm": lambda: arrays["tc"].shape[2], "ttv_front": lambda: arrays["ids"].shape[1],
             "ttv_gen": lambda: arrays["x_frame"].shape[2]}.get(
            kind, lambda: arrays["x"].shape[2])()
        mask = Fh.sequence_mask(d("lengths"), T)
    with torch.no_grad():
        if kind in ("act1d", "amp_block", "convtr", "dblock", "speechsr", "speechsr_real"):
            out = [mod(d("x"))]
        elif kind == "wn":
            out = [mod(d("x"), mask, g=d("g"))]
        elif kind == "rel_mha":
            out = [mod(d("x"), d("x"), mask_q=mask, mask_k=mask)]
        elif kind == "vits_encoder":
            out = [mod(d("x"), mask)]
        elif kind == "tts_e2e":
            from megatts2_hierspeechpp_amd.inference_plm import tts
            n, tm = arrays["ids"].shape[1], arrays["src_mel"].shape[2]
            dl = lambda v: torch.tensor(v, dtype=torch.int64, device=device)
            wav, audio = tts(mod, d("ids"), dl([n]), d("tone"), d("language"), d("mel_ttv"),
                             dl([arrays["mel_ttv"].shape[2]]), d("src_mel"), dl([tm, tm]),
                             noise_scale_vc=meta["noise_scale"], denoise_ratio=meta["denoise_ratio"], noise=d("noise"),
                             return_float=True)
            out = [audio, wav.float().reshape(-1)]
        elif kind == "ttv_front":
            xf, g, fl, _ = mod.inf_extract_tc_latent(d("ids"), d("lengths"), d("mel"), d("mel_lengths"), d("tone"),
                                                     d("language"))
            out = [xf, g, fl]
        elif kind == "w2v":
            out = [mod(d("x").unsqueeze(1))]
        elif kind == "mp_conformer":
            # the reference holds [A, N, C]; the product module is channel-major [A, C, N]
            out = [mod(d("x").transpose(1, 2).contiguous()).transpose(1, 2)]
        elif kind == "mp_dense_encoder":
            out = [mod(d("x"))]
        elif kind == "denoise":
            # The network and the inverse STFT against the reference, on the spectrogram the REFERENCE run saw (stored
            # in the fixture): the first and last frame of a centred, reflect-padded STFT are even-symmetric, their
            # spectrum is real up to rounding, and the phase of a negative real number with noise for an imaginary part
            # is +pi or -pi by the accident of the FFT's summation order -- a 2 pi jump in an input FEATURE of the
            # network that no other DFT implementation (not even torch.stft on another CPU) reproduces.  The product
            # STFT itself is compared on the continuous quantities in test_denoiser_stft; the whole product call in
            # test_denoise_end_to_end.
            from megatts2_hierspeechpp_amd.denoiser.infer import mag_pha_istft
            w = torch.from_numpy(arrays["wav"])
            norm = torch.sqrt(len(w) / torch.sum(w ** 2.0))
            amp_in, pha_in = d("amp_in"), d("pha_in")
            amp_g, pha_g, _ = mod(amp_in, pha_in)
            out = [mag_pha_istft(amp_g, pha_g, 400, 100, 400, 0.3, scale=1.0 / float(norm)), amp_g]
        elif kind == "ttv_infer":
            n = arrays["ids"].shape[1]
            dl = lambda v: torch.tensor(v, dtype=torch.int64, device=device)
            out = list(mod.infer(d("ids"), dl([n]), d("mel"), dl([arrays["mel"].shape[2]]), d("tone"), d("language"),
                                 dur=d("dur")))
        elif kind == "vc_plain":
            out = [mod.voice_conversion(d("w2v"), d("src_length"), d("mel"), d("trg_length"), d("f0"),
                                        noise_scale=meta["noise_scale"], noise=d("noise"),
                                        uncond=bool(meta.get("uncond", False)))]
        elif kind == "ttv_gen":
            out = list(mod.inf_plm_gen(d("x_frame"), d("g"), d("codes"), d("frame_lengths"), None))
        elif kind == "plm":
            # one batched run; rows shorter than the longest are cut to the fixture's -1 / 0 padding
            codes, logits = mod.infer(d("tc"), return_logits=True)
            valid = mask[:, 0] > 0
            out = [torch.where(valid, codes.float(), torch.full_like(valid, -1.0, dtype=torch.float32)),
                   logits * valid.unsqueeze(-1)]
        elif kind == "dit_block":
            out = [mod(d("x").transpose(1, 2).contiguous(), d("c"), mask).transpose(1, 2)]
        elif kind == "coupling":
            out = [mod(d("x"), mask, g=d("c"), reverse=True)]
        elif kind == "flow":
            out = [mod(d("x"), mask, g=d("g"), reverse=True)]
        elif kind == "style_encoder":
            out = [mod(d("mel"), mask)]
        elif kind == "posterior_sf":
            out = list(mod(d("w2v"), d("f0"), mask, g=d("g"), noise=d("noise")))
        elif kind == "source_network":
            out = list(mod(d("z"), d("g")))
        elif kind == "generator":
            out = [mod(d("z"), d("e"), g=d("g"))]
        elif kind == "infer":
            out = list(mod.infer(d("mel"), d("w2v"), d("lengths"), d("f0"), noise=d("noise")))
        elif kind == "vc":
            out = [mod.voice_conversion_noise_control(d("w2v"), d("src_length"), d("mel"), d("trg_length"), d("f0"),
                                                      noise_scale=meta["noise_scale"],
                                                      denoise_ratio=meta["denoise_ratio"], noise=d("noise"))]
        else:
            raise KeyError(kind)
    torch.cuda.synchronize()
    return [o.detach().cpu().numpy() for o in out]
