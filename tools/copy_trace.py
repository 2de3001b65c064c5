#!/usr/bin/env python3
"""Where do the device-to-device copies of a vocoder step (`__amd_rocclr_copyBuffer` in the kernel stats) come from?
One eager 32 x 4 s step under torch.profiler with Python stacks: every aten::copy_ (and cat / clone / contiguous that ends in
one) attributed to the innermost frame inside this package.      python tools/copy_trace.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = bench.parse_args([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.VocoderWorkload(args, 0, 1, dev)
wl.model.finalize(dev)
wl.prepare(0, args.batch)
wl.eager_step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    wl.eager_step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_",):
        where = "?"
        for fr in ev.stack:
            if "megatts2_hierspeechpp_amd" in fr:
                where = fr.split("megatts2_hierspeechpp_amd/")[-1]
                break
        cnt[where] += 1
print(sum(cnt.values()), "aten::copy_ calls in one step")
for k, v in cnt.most_common(40):
    print(v, k)
