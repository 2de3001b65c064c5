#!/usr/bin/env python3
"""One warm-up + one measured EAGER pass of the dominant stage of an extra config, for `rocprofv3 --pmc` (graph-replayed
kernels are invisible to the profiler; tools/pmc_traffic_extra.sh wraps this):
    tts   BASELINE.json configs[2]: the PLM greedy loop on the x_frame of the batch-16 TTS workload (200 steps, eager)
    sr48  BASELINE.json configs[3]: the SpeechSR48 stage on the vocoder's 32 x 4 s output
    vc_w2v    extra_configs.vc_b1_4s: the wav2vec2 producer on 4 s of padded source audio
    denoiser  extra_configs.tts_prompt_denoise: the MP-SENet denoiser on the padded 3-s prompt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import bench_extra as BE  # noqa: E402
from megatts2_hierspeechpp_amd import hierspeechpp_speechsynthesizer as hss  # noqa: E402
from megatts2_hierspeechpp_amd import synth  # noqa: E402

which = sys.argv[1]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hss.SERIAL_STREAMS = True
if which == "tts":
    from megatts2_hierspeechpp_amd.ttv_v1.t2w2v_transformer import Megatts2PLM1
    plm = Megatts2PLM1()
    plm.load_state_dict({k: torch.from_numpy(synth.synth_tensor("plm." + k, tuple(v.shape), 0)) for k, v in plm.state_dict().items()})
    plm.finalize(dev)
    tc = torch.from_numpy(np.random.default_rng(1).standard_normal((16, 256, 200)).astype(np.float32)).to(dev)
    for _ in range(2):
        plm.infer(tc)
    torch.cuda.synchronize()
elif which == "sr48":
    from megatts2_hierspeechpp_amd.hip_layers import finalize
    from megatts2_hierspeechpp_amd.speechsr48k.speechsr import SynthesizerTrn as SpeechSR
    sr = SpeechSR(128, 30, "0", [3, 7, 11], [[1, 3, 5]] * 3, [3], 32, [3])
    sr.load_state_dict({k: torch.from_numpy(synth.synth_tensor("sr." + k, tuple(v.shape), 0)) for k, v in sr.state_dict().items()})
    finalize(sr, dev)
    o = torch.tanh(torch.randn(32, 1, 64000, device=dev))
    for _ in range(2):
        sr(o)
    torch.cuda.synchronize()
elif which == "vc_w2v":
    from megatts2_hierspeechpp_amd import functional as Fh, inference_vc as IV
    from megatts2_hierspeechpp_amd.extract_w2v import Wav2vec2
    from megatts2_hierspeechpp_amd.hip_layers import finalize
    w2v = Wav2vec2(layer=7)
    w2v.load_state_dict({k: torch.from_numpy(synth.synth_tensor("w2v." + k, tuple(v.shape), 0)) for k, v in w2v.state_dict().items()})
    finalize(w2v, dev)
    src = IV.pad_source(torch.from_numpy(BE._speechlike(64000 - 640, 11)).to(dev))
    for _ in range(2):
        w2v(Fh.reflect_pad(src, 40))
    torch.cuda.synchronize()
elif which == "denoiser":
    import types
    from megatts2_hierspeechpp_amd.denoiser.generator import MPNet
    from megatts2_hierspeechpp_amd.denoiser.infer import denoise
    hd = types.SimpleNamespace(dense_channel=64, compress_factor=0.3, num_tsconformers=4, beta=2.0, sampling_rate=16000,
                               n_fft=400, hop_size=100, win_size=400)
    den = MPNet(hd)
    den.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 7)) for k, v in den.state_dict().items()})
    den.finalize(dev)
    padded = torch.zeros(49600, device=dev)
    padded[:48000].copy_(torch.from_numpy(BE._speechlike(48000, 21)).to(dev)[0])
    for _ in range(2):
        denoise(padded, den, hd)
    torch.cuda.synchronize()
else:
    raise SystemExit("tts | sr48 | vc_w2v | denoiser")
print("done", which)
