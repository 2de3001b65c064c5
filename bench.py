#!/usr/bin/env python3
"""Headline benchmark: HierSpeech++ vocoder path, batch 32 x 4 s per GPU (BASELINE.json
configs[1]), 16 kHz output samples / s over the whole job + RTF.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under a launcher (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`, one rank per GPU over RCCL) and on its own --
with WORLD_SIZE unset the parent process starts that launcher as a child BEFORE anything touches the GPU
(never a re-exec), relays rank 0's JSON line and exits with the child's status.

A step = one ``SynthesizerTrn.infer`` pass (style encoder -> SF prior encoder -> two reversed DiT coupling
flows -> source network -> BigVGAN-style generator) over this rank's shard of a global batch of 32 x N
utterances (parallel.shard_range: contiguous blocks, no data-path collective), synthetic (mel, w2v, f0) already
resident in HBM; weights are the synthetic recipe (no checkpoints exist offline).  Rank 0 packs the weights,
every other rank only lays the arena out and receives the bytes by ONE chunked RCCL broadcast.

Printed JSON (one line, rank 0): metric/value per the driver contract plus
  pack_ms (rank 0: fold + pack + upload of the weight arena), broadcast_ms (the dist.broadcast loop alone,
  barrier-bracketed; null at N = 1), broadcast_gbs, rank_ms_per_step {min, max}, rccl_world -- the multi-GPU side
  event_median_ms -- median of the K steps, each bracketed by a HIP-event pair (SURVEY.md 8d protocol)
  roofline      -- dominant kernel (conv1d_mfma_kernel): algorithmic FLOP of all its launches in one step /
                   their summed duration, timed live with events on the launch stream, against the
                   157.3 TFLOP/s fp32 MFMA peak; `traffic` from the committed PMC profile when it was
                   taken on this very build and launch mix, else null with the reason
  cpu_baseline  -- the CPU oracle (oracle/hsp_oracle.py) on this box's host cores: configs[0] (1 x 1 s) and a
                   bounded sample of configs[1] (8 x 4 s), N=1 only
  extra_configs -- configs[0] on the GPU (1 x 1 s latency), configs[2] (full text->wav, batch 16), configs[3]
                   (vocoder + SpeechSR48, batch 32) and the batch-1 latencies of the reference's own usage
                   (vocoder 1 x 4 s, full TTS 1 x 4 s), N=1 only (tools/bench_extra.py)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector peak
HBM_PEAK_GBS = 8000.0
TRAFFIC_PROFILE = "r06_traffic.json"   # tools/pmc_traffic.sh + tools/pmc_summarize.py; quoted only when it matches this build
METRIC = "16kHz audio samples/sec (whole node) + RTF, HierSpeech++ vocoder batch=32"

VOC_CFG = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3,
               p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
               upsample_rates=[4, 5, 4, 2, 2], upsample_initial_channel=1024, upsample_kernel_sizes=[8, 11, 8, 4, 4],
               gin_channels=256)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (global batch = batch x gpus)")
    ap.add_argument("--seconds", type=float, default=4.0, help="audio seconds per utterance")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="utterances of the whole job (default batch x gpus); need not divide by the rank count: "
                         "parallel.shard_range hands out contiguous shards whose sizes differ by at most one")
    ap.add_argument("--force-dist", action="store_true",
                    help="form the RCCL process group and run every collective of the N > 1 path (weight broadcast, "
                         "barriers, timing reductions) at world size 1 as well")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs (configs[2] and configs[3])")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--dump-launches", default=None, help="write a per-shape table of the conv launches here")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------ self-launch
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_children(argv, gpus):
    """`python bench.py --gpus N` without a launcher: start `torch.distributed.run` as a CHILD process (this
    parent has not imported torch and never touches the GPU), relay its output, return its exit status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no result line\n")
        rc = 1
    if line is not None:
        print(line, flush=True)
    return rc


# ------------------------------------------------------------------------ workload
class VocoderWorkload:
    """configs[1]: SynthesizerTrn.infer on this rank's utterances."""

    def __init__(self, args, rank, world, dev):
        import torch
        from megatts2_hierspeechpp_amd import synth
        from megatts2_hierspeechpp_amd.hierspeechpp_speechsynthesizer import SynthesizerTrn
        self.args, self.rank, self.world, self.dev = args, rank, world, dev
        self.frames = int(round(args.seconds * 50))
        self.model = SynthesizerTrn(641, 61440 // 320, **VOC_CFG)
        self.sd_np = None
        if rank == 0:  # only rank 0 materialises weights; the others receive the packed arena
            self.sd_np = {k: synth.synth_tensor(k, tuple(v.shape), 0) for k, v in self.model.state_dict().items()}
            self.model.load_state_dict({k: torch.from_numpy(v) for k, v in self.sd_np.items()})
        self.samples_per_utterance = 320 * self.frames

    def prepare(self, lo, hi):
        import torch
        from megatts2_hierspeechpp_amd import synth
        self.B = hi - lo
        inp = synth.synth_inputs(self.B, self.frames, seed=20240 + lo)
        self.inp = {k: torch.from_numpy(v).to(self.dev) for k, v in inp.items()}

    def eager_step(self):
        d = self.inp
        return self.model.infer(d["mel"], d["w2v"], d["length"], d["f0"], noise=d["noise"])

    def make_step(self):
        """The whole forward (~800 launches) is captured once into a hipGraph and replayed: every launch of a
        step still executes, only the host-side submission cost goes."""
        import torch
        self.eager_step()  # also raises the kernels' dynamic-LDS limits outside the capture
        torch.cuda.synchronize()
        if self.args.no_graph:
            return self.eager_step
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = self.eager_step()

        def step():
            graph.replay()
            return static_out
        return step

    def check(self, out):
        import torch
        o = out[0]
        assert o.shape == (self.B, 1, 320 * self.frames) and bool(torch.isfinite(o).all())

    def describe(self, world):
        a = self.args
        return {"workload": f"vocoder-only infer(): {a.batch} utterances x {a.seconds:g} s per GPU "
                            f"(BASELINE.json configs[1]), synthetic weights", "batch_per_gpu": a.batch,
                "frames": self.frames, "global_batch": a.batch * world,
                "parallelism": f"dp{world} (contiguous utterance shards of the global batch, one RCCL weight broadcast)",
                "launch_mode": "eager" if a.no_graph else "hipGraph replay of the captured step"}


def run_bench(args, make_workload, backend="nccl", device=None):
    """The control flow of one rank (N = 1 included).  `make_workload(args, rank, world, dev)` returns an object
    with .model (has .finalize(device, materialize)), .prepare(lo, hi), .make_step(), .check(out),
    .samples_per_utterance and .describe(world).  Returns (result dict or None on ranks != 0, workload)."""
    import torch
    import torch.distributed as dist
    from megatts2_hierspeechpp_amd import parallel

    rank, local_rank, world = parallel.init_distributed(backend, force=getattr(args, "force_dist", False))
    coll = parallel.collectives_on()      # N > 1, or --force-dist: the collectives of the multi-GPU path run
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dev = device if device is not None else torch.device("cuda", local_rank)
    on_gpu = dev.type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    if on_gpu:
        torch.cuda.set_device(dev)

    wl = make_workload(args, rank, world, dev)
    # weights: rank 0 folds + packs, everyone else lays the same arena out; one chunked broadcast
    sync()
    if coll:
        dist.barrier()
    tm = {}
    arena = parallel.finalize_distributed(wl.model, dev, src=0, timings=tm)
    sync()
    # pack_ms: rank 0's fold + pack + upload (the other ranks only lay the arena out); broadcast_ms: the
    # dist.broadcast loop alone, max over ranks, None when no collective ran (world size 1)
    pack_ms = parallel.gather_floats(tm["pack_ms"], dev)[0]
    broadcast_ms = parallel.barrier_max(tm["broadcast_ms"], dev) if tm["broadcast_ms"] is not None else None
    # derive_ms: every rank's own derivation of the frequency-domain matrices from the taps it holds (max over ranks)
    derive_ms = parallel.barrier_max(tm["derive_ms"], dev) if tm.get("derive_ms") is not None else None

    n_utt = args.global_batch if getattr(args, "global_batch", None) else args.batch * world
    lo, hi = parallel.shard_range(n_utt, rank, world)
    wl.prepare(lo, hi)
    step = wl.make_step()

    for _ in range(args.warmup):
        step()
    sync()
    if coll:
        dist.barrier()
    sync()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out = step()
        if on_gpu:
            e1.record()
            events.append((e0, e1))
    sync()
    local = time.perf_counter() - t0
    if coll:
        dist.barrier()
    sync()
    elapsed = parallel.barrier_max(time.perf_counter() - t0, dev)
    per_rank = parallel.gather_floats(1e3 * local / args.steps, dev)
    wl.check(out)

    samples_per_step = n_utt * wl.samples_per_utterance
    audio_s_per_step = samples_per_step / 16000.0
    result = None
    if rank == 0:
        cfg = wl.describe(world)
        cfg["weights_mb"] = arena.buffer.numel() * 4 / 1e6     # what travels (the folded weights); derived data is below
        cfg["shard_of_rank0"] = [lo, hi]
        cfg["global_batch"] = n_utt
        result = {
            "metric": METRIC, "value": samples_per_step * args.steps / elapsed, "unit": "samples/s",
            "rtf": (elapsed / args.steps) / audio_s_per_step, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if getattr(args, "global_batch", None) else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": cfg,
            "pack_ms": pack_ms, "broadcast_ms": broadcast_ms, "broadcast_mb": tm.get("broadcast_mb"),
            "derive_ms": derive_ms, "derived_mb": tm.get("derived_mb"),
            "broadcast_gbs": (arena.buffer.numel() * 4 / 1e9) / (broadcast_ms * 1e-3) if broadcast_ms else None,
            "rank_ms_per_step": {"min": min(per_rank), "max": max(per_rank)},
            "rccl_world": dist.get_world_size() if dist.is_initialized() else 1, "backend": backend if coll else None,
        }
        if events:
            import numpy as np
            result["event_median_ms"] = float(np.median([a.elapsed_time(b) for a, b in events]))
    return result, wl


# ------------------------------------------------------------------------ roofline (vocoder, live)
def _build_id():
    """what the committed PMC traffic file must match: the hash of the kernel sources the library is built from"""
    from megatts2_hierspeechpp_amd.build import source_id
    return source_id()


def vocoder_roofline(args, wl, result):
    import ctypes as C
    import torch
    from megatts2_hierspeechpp_amd import _lib as L
    from megatts2_hierspeechpp_amd import functional as Fh
    from megatts2_hierspeechpp_amd import hierspeechpp_speechsynthesizer as hss
    from megatts2_hierspeechpp_amd import hip_layers
    # launches must run back to back on one stream here: with the AMP chains / batch groups on
    # side streams the event-bracketed durations of concurrent kernels would overlap
    # (round 3) exactly the launches of the timed step -- four front groups, the same tile-count policy for the fused
    # WN / FFN entry points -- only issued serially
    saved = hss.SERIAL_STREAMS
    hss.SERIAL_STREAMS = True
    rec, act_rec = [], []

    tiles = {}

    def hook(kind, fl, nb, e0, e1, la):
        # three kernels sit behind hsp_conv1d_mfma_f32: ask the library which one this launch took
        if kind == "hsp_conv1d_mfma_f32":
            plan = (C.c_int32 * 4)()
            L.check(L.lib().hsp_conv1d_mfma_plan(C.byref(la), C.byref(plan)), "hsp_conv1d_mfma_plan")
            tiles[len(rec)] = f"{plan[0]}x{plan[1]}"
            kind = "hsp_conv1d_mfma_f32" if plan[2] > 0 else (
                "hsp_conv1d_mfma_f32/rgemm" if plan[2] == -1 else "hsp_conv1d_mfma_f32/bgemm")
        rec.append((kind, fl, nb, e0, e1, la if isinstance(la, int) else
                    (la.Cin, la.Cout, la.K, la.dil, la.Lout, la.prologue, la.rows, int(bool(la.w_bs))) if la is not None
                    else (0, 0, 0, 0, 0, 0, 0, 0)))

    Fh.ACT_HOOK = lambda nb, e0, e1: act_rec.append((nb, e0, e1))
    hip_layers.LAUNCH_HOOK = hook
    try:
        wl.eager_step()
        torch.cuda.synchronize()
    finally:
        hip_layers.LAUNCH_HOOK = None
        Fh.ACT_HOOK = None
        hss.SERIAL_STREAMS = saved
    mf = [(fl, nb, e0.elapsed_time(e1)) for kind, fl, nb, e0, e1, _ in rec if kind == "hsp_conv1d_mfma_f32"]
    tg = [(fl, nb, e0.elapsed_time(e1)) for kind, fl, nb, e0, e1, _ in rec
          if kind in ("hsp_conv1d_mfma_f32/rgemm", "hsp_conv1d_mfma_f32/bgemm")]
    if args.dump_launches:
        agg = {}
        for kind, fl, nb, e0, e1, shp in rec:
            k = (kind,) + (shp[:7] if isinstance(shp, tuple) else (0, 0, 0, 0, 0, 0, 0))
            n, f, m = agg.get(k, (0, 0, 0.0))
            agg[k] = (n + 1, f + fl, m + e0.elapsed_time(e1))
        with open(args.dump_launches, "w") as fh:
            fh.write("kind Cin Cout K dil Lout prologue rows | launches gflop ms TF/s\n")
            for k, (n, f, m) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
                fh.write(f"{k[0][4:].replace('_f32', ''):14s} {k[1]:5d} {k[2]:5d} {k[3]:3d} {k[4]:2d} {k[5]:6d} {k[6]} {k[7]} | "
                         f"{n:4d} {f / 1e9:10.1f} {m:9.3f} {f / (m * 1e-3) / 1e12:7.2f}\n")
    tot_ms = sum(m for _, _, m in mf)
    tot_fl = sum(f for f, _, _ in mf)
    tot_b = sum(b for _, b, _ in mf)
    ach = tot_fl / (tot_ms * 1e-3) / 1e12
    # the same launches by tile shape: the Generator's long-sequence convs (128x128, 64x256, 32x512) against the
    # short-sequence shapes of the four 8-utterance front groups (64x64, 64x128-gated, 32x128)
    by_tile = {}
    for i, (kind, fl, nb, e0, e1, _) in enumerate(rec):
        if kind == "hsp_conv1d_mfma_f32":
            t = by_tile.setdefault(tiles.get(i, "?"), [0, 0.0, 0.0])
            t[0] += 1
            t[1] += fl
            t[2] += e0.elapsed_time(e1)
    by_tile = {k: {"launches": n, "gflop": f / 1e9, "ms": m, "tflops": f / (m * 1e-3) / 1e12,
                   "frac": f / (m * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS} for k, (n, f, m) in sorted(by_tile.items())}

    # HBM bytes of the same kernel from the PMC counters: measured by tools/pmc_traffic.sh (rocprofv3 cannot wrap
    # a run from inside) and committed under profiles/.  Quoted only when that profile was taken on THIS build of
    # the library, with this workload and this launch mix; otherwise null, with the reason.
    traffic, traffic_note, tj = None, None, None
    path = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
    if not os.path.exists(path):
        traffic_note = "no profiles/" + TRAFFIC_PROFILE
    else:
        with open(path) as fh:
            tj = json.load(fh)
        if tj.get("kernel_source_sha16") != _build_id():
            traffic_note, tj = f"profile taken on kernel sources {tj.get('kernel_source_sha16')}, this run uses {_build_id()}", None
        elif (tj.get("batch_per_gpu"), tj.get("frames")) != (wl.B, wl.frames):
            traffic_note, tj = "profile taken on another workload size", None
        elif tj.get("conv1d_mfma_launches_per_step") != len(mf) or tj.get("act1d_launches_per_step") != len(act_rec) or \
                tj.get("cprod3_kernel_launches_per_step", 0) != sum(1 for r_ in rec if r_[0] == "hsp_cprod3_f32"):
            traffic_note, tj = "profile taken with another launch mix", None
        else:
            traffic = tj["conv1d_mfma_bytes_per_step"]["calibrated"] / len(mf)
            traffic_note = ("HBM bytes per launch: PMC FETCH_SIZE and WRITE_SIZE (separate passes), FETCH scaled by the "
                            "factor calibrated on a float4 copy of known size; raw / x2 / calibrated per step: "
                            + json.dumps(tj["conv1d_mfma_bytes_per_step"]))
    result["roofline"] = {
        "kernel": "conv1d_mfma_kernel (every tile shape the timed step launches; the 1x1 token-GEMM launches of the same "
                  "entry point -- bgemm_kernel / rgemm_kernel -- are excluded: %d launches, %.2f ms per step)"
                  % (len(tg), sum(m for _, _, m in tg)),
        "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
        "traffic_source": "committed PMC profile profiles/" + TRAFFIC_PROFILE + " (tools/pmc_traffic.sh; rocprofv3 cannot wrap a run "
                          "from inside), quoted only when its kernel-source hash, workload and launch mix equal this run's -- "
                          "not measured by this run",
        "algorithmic_bytes_per_launch": tot_b / max(len(mf), 1),
        "launches_per_step": len(mf), "avg_launch_ms": tot_ms / max(len(mf), 1), "kernel_ms_per_step": tot_ms,
        "algorithmic_gflop_per_step": tot_fl / 1e9, "algorithmic_mb_per_step": tot_b / 1e6,
        "hbm_frac_of_8TBs": (tot_b / (tot_ms * 1e-3) / 1e9) / HBM_PEAK_GBS,
        "share_of_step_time_single_stream": tot_ms / result["ms_per_step"], "by_tile_shape": by_tile,
        "timing": "one extra step, the timed step's own launches serialised on one stream, event pair per launch",
    }
    if tj:
        # (round 5) the whole step's HBM bytes from the same committed PMC passes, class by class, against SURVEY.md 8(d)'s
        # algorithmic bytes of the full infer path: which kernel owns the waste
        result["roofline"]["step_traffic_gb"] = tj.get("step_traffic_gb")
        result["roofline"]["step_algorithmic_gb"] = tj.get("step_algorithmic_gb")
        result["roofline"]["step_traffic_by_class_gb"] = tj.get("step_traffic_by_class_gb")
        result["roofline"]["traffic_measured_over_algorithmic"] = tj.get("measured_over_algorithmic")
    # whole-step fractions SURVEY.md 8(d) defines (Generator-only algorithmic work per audio-second over the step time)
    audio_s = wl.B * args.seconds          # rank 0's shard: what the timed step of THIS rank synthesised (--global-batch)
    result["roofline"]["step_fma_fraction"] = 63.3e9 * audio_s / (result["ms_per_step"] * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * 1e12)
    result["roofline"]["step_hbm_fraction"] = 391e6 * audio_s / (result["ms_per_step"] * 1e-3) / (HBM_PEAK_GBS * 1e9)
    # continuity with BENCH_r02 / r03 (`round2_launch_mix`): the same figure over the launches with more than 200 output
    # columns per utterance -- the Generator's and SourceNetwork's convs, round 2's 167-launch population (its per-launch
    # pass ran the 50 Hz front part in a fused kernel that round 4 retired, so the population is selected by shape now)
    long_ = [(fl, e0.elapsed_time(e1)) for kind, fl, nb, e0, e1, shp in rec if kind == "hsp_conv1d_mfma_f32" and shp[4] > 200]
    ms2, fl2 = sum(m for _, m in long_), sum(f for f, _ in long_)
    result["roofline"]["round2_launch_mix"] = {
        "launches_per_step": len(long_), "kernel_ms_per_step": ms2, "achieved": fl2 / (ms2 * 1e-3) / 1e12,
        "frac": fl2 / (ms2 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
        "note": "conv1d_mfma_kernel launches with more than 200 output columns per utterance (Generator + SourceNetwork): "
                "the population BENCH_r02's roofline.frac was computed over; a subset of `roofline`'s launches"}
    # (round 4) the same population split by what the launch is: a conv of the path in its direct form, or the K = 1
    # channel product of a frequency-domain conv (per-bin weights, hsp_conv1d_args.w_bs) -- a launch class round 3 did not have
    # (round 5) the channel products run on their own kernel (cprod3_kernel, hsp_cprod3_f32: three real C x C products per
    # bin, 6 C^2 flops per column instead of the block form's 8 C^2), so `roofline` -- conv1d_mfma_kernel -- now holds the
    # direct convs only (unless HSP_FFT_PRODUCT=block); `mfma_gemm_population` is the union, round 4's population.
    cp3 = [(fl, nb, e0.elapsed_time(e1)) for kind, fl, nb, e0, e1, _ in rec if kind == "hsp_cprod3_f32"]
    for name, part in (("direct_convs", [(fl, e0.elapsed_time(e1)) for kind, fl, nb, e0, e1, shp in rec
                                         if kind == "hsp_conv1d_mfma_f32" and shp[7] == 0]),
                       ("channel_products", [(fl, e0.elapsed_time(e1)) for kind, fl, nb, e0, e1, shp in rec
                                             if kind == "hsp_conv1d_mfma_f32" and shp[7] == 1] + [(f, m) for f, _, m in cp3])):
        if part:
            pms, pfl = sum(m for _, m in part), sum(f for f, _ in part)
            result["roofline"][name] = {"launches_per_step": len(part), "kernel_ms_per_step": pms, "achieved": pfl / (pms * 1e-3) / 1e12,
                                        "frac": pfl / (pms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    if cp3:
        result["roofline"]["channel_products"].update({
            "kernel": "cprod3_kernel (hsp_cprod3_f32)", "flops": "executed: 2 x 3 C^2 x Np x 64 bins per launch",
            "algorithmic_mb_per_step": sum(b for _, b, _ in cp3) / 1e6,
            "block_form_equivalent_tflops": sum(f for f, _, _ in cp3) * (4.0 / 3.0) / (sum(m for _, _, m in cp3) * 1e-3) / 1e12})
        ums, ufl = tot_ms + sum(m for _, _, m in cp3), tot_fl + sum(f for f, _, _ in cp3)
        result["roofline"]["mfma_gemm_population"] = {
            "launches_per_step": len(mf) + len(cp3), "kernel_ms_per_step": ums, "achieved": ufl / (ums * 1e-3) / 1e12,
            "frac": ufl / (ums * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "note": "conv1d_mfma_kernel + cprod3_kernel launches, executed flops: the population BENCH_r04's roofline.frac "
                    "(0.573) was computed over, when the channel products still ran on conv1d_mfma_kernel"}
    # (round 4) the long AMP convs run in their frequency-domain form (forward DFT, ONE batched 1x1 product over the 64
    # bins on conv1d_mfma_kernel, inverse DFT: csrc/hsp_dftseg.hip).  `roofline` above counts the product launches with
    # the flops they EXECUTE; here the convs they stand for: the direct form's algorithmic flops over the time of all
    # three launches -- the rate the direct kernel would have to reach to tie (its peak is 157.3).
    fc = [(fl, nb, e0.elapsed_time(e1)) for kind, fl, nb, e0, e1, _ in rec if kind == "hsp_fftconv"]
    n_fc = sum((la if isinstance(la, int) else 1) for kind, fl, nb, e0, e1, la in rec if kind == "hsp_fftconv")   # a fused pair is one record
    if fc:
        tr = [(kind, fl, e0.elapsed_time(e1)) for kind, fl, nb, e0, e1, _ in rec if kind.startswith("hsp_dftseg")]
        f_ms, f_fl = sum(m for _, _, m in fc), sum(f for f, _, _ in fc)
        result["roofline"]["frequency_domain_convs"] = {
            "convs_per_step": n_fc, "ms_per_step_all_three_launches": f_ms, "algorithmic_gflop_per_step": f_fl / 1e9,
            "algorithmic_tflops": f_fl / (f_ms * 1e-3) / 1e12,
            "transform_launches_per_step": len(tr), "transform_ms_per_step": sum(m for _, _, m in tr),
            "transform_executed_tflops": sum(f for _, f, _ in tr) / (sum(m for _, _, m in tr) * 1e-3) / 1e12,
            "note": "algorithmic flops of the direct convs (2 k C^2 per output) over the time of forward DFT + channel product + "
                    "inverse DFT (an AMP pair: forward, product, inverse + activation + forward in one launch, product, "
                    "inverse); the product launches are also in `roofline` with their executed flops"}
    if act_rec:
        # second kernel of the step by time: the stand-alone anti-aliased activation, HBM-bound by design
        a_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in act_rec)
        a_b = sum(nb for nb, _, _ in act_rec)
        a_gbs = a_b / (a_ms * 1e-3) / 1e9
        result["roofline_activation"] = {
            "kernel": "act1d_seg_kernel (hsp_act1d_snakebeta_f32)", "bound": "hbm", "achieved": a_gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_gbs / HBM_PEAK_GBS, "launches_per_step": len(act_rec),
            "kernel_ms_per_step": a_ms, "algorithmic_mb_per_step": a_b / 1e6,
            "algorithmic_bytes": "one fp32 read + one fp32 write per element",
            "algorithmic_bytes_per_launch": a_b / len(act_rec),
            "traffic": (tj["act1d_seg_bytes_per_step"]["calibrated"] / len(act_rec)) if tj else None,
            "timing": "same pass as `roofline`",
        }


def _progress(msg):
    """stderr only (stdout carries the one JSON line): a long default run keeps saying what it is doing"""
    sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] {msg}\n")
    sys.stderr.flush()


# ------------------------------------------------------------------------ CPU baseline
def _cpu_model_string():
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for ln in out.splitlines():
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except (OSError, subprocess.SubprocessError):
        pass
    return "unknown"


def cpu_baseline(args, wl, result):
    """The oracle on this box's host cores, same synthetic inputs: configs[0] (1 x 1 s) and a bounded sample of
    configs[1] (8 x 4 s instead of 32 x 4 s: SURVEY.md 8d allows the cut), plus the GPU path's error on the
    very utterances the oracle synthesised."""
    import numpy as np
    import torch
    from megatts2_hierspeechpp_amd import synth
    from oracle import hsp_oracle as O
    host_cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sd = {k: torch.from_numpy(v) for k, v in wl.sd_np.items()}

    def cpu_run(batch, frames):
        ci_ = synth.synth_inputs(batch, frames, seed=20240)
        t_ = lambda k: torch.from_numpy(ci_[k])
        c0 = time.perf_counter()
        with torch.no_grad():
            out, _ = O.synth_infer(sd, VOC_CFG, t_("mel"), t_("w2v"), t_("length"), t_("f0"), t_("noise"))
        return time.perf_counter() - c0, ci_, out

    def gpu_err(ci, ro):
        with torch.no_grad():
            g_ = lambda k: torch.from_numpy(ci[k]).to(wl.dev)
            go, _ = wl.model.infer(g_("mel"), g_("w2v"), g_("length"), g_("f0"), noise=g_("noise"))
        return float((go.cpu() - ro).abs().max())

    # The thread count is MEASURED, not assumed (SURVEY.md 8d names os.cpu_count(); on the 256-thread host of the GPU box
    # every hardware thread oversubscribes oneDNN's conv threading -- a 2 x 4 s sample did not finish in 7 minutes, rounds
    # 2-3): the 1 x 1 s sample of configs[0] at 8 ... 128 threads, second of two runs each (the sweep stops at the first
    # count whose run takes longer than 8 s); the baseline then runs at the count that was fastest.
    torch.set_num_threads(min(host_cpus, 8))
    cpu_run(1, 25)                          # warm-up (thread pools, oneDNN primitives)
    scaling = []
    for n in (8, 16, 32, 64, 128):
        if n > host_cpus and scaling:
            break
        torch.set_num_threads(min(n, host_cpus))
        cpu_run(1, 50)
        tt = cpu_run(1, 50)[0]
        scaling.append({"threads": min(n, host_cpus), "seconds": tt, "value": 320 * 50 / tt})
        _progress(f"cpu_baseline: 1 x 1 s at {min(n, host_cpus)} threads: {tt:.2f} s")
        if tt > 8.0:
            break
    cores = min(scaling, key=lambda s_: s_["seconds"])["threads"]
    # a larger sample may scale further: the fastest count and twice it, once each on 2 x 4 s
    pick = {}
    for n in sorted({cores, min(2 * cores, host_cpus)}):
        torch.set_num_threads(n)
        pick[n] = cpu_run(2, wl.frames)[0]
        _progress(f"cpu_baseline: 2 x {wl.frames / 50:g} s at {n} threads: {pick[n]:.2f} s")
    cores = min(pick, key=pick.get)
    torch.set_num_threads(cores)
    t1 = []
    for _ in range(3):
        tt, ci1, ro1 = cpu_run(1, 50)
        t1.append(tt)
    m1 = float(np.median(t1))
    # configs[1] sample: 8 x 4 s if one run fits ~25 s of CPU work, else 1 x 4 s; median of three runs
    est8 = m1 * 4 * 8
    sb = 8 if est8 < 25.0 else 1
    t8s = []
    for i in range(3):
        _progress(f"cpu_baseline: {sb} x {wl.frames / 50:g} s at {cores} threads, run {i + 1} of 3")
        tt, ci8, ro8 = cpu_run(sb, wl.frames)
        t8s.append(tt)
    t8 = float(np.median(t8s))
    result["cpu_baseline"] = {
        "value": sb * 320 * wl.frames / t8, "unit": "samples/s", "cores": cores, "kind": "port",
        "sample": f"oracle synth_infer on {sb} x {wl.frames / 50:g} s of configs[1]'s 32 x {wl.frames / 50:g} s "
                  f"(median of 3 runs: {', '.join(f'{t:.2f}' for t in t8s)} s)", "rtf": t8 / (sb * wl.frames / 50),
        "thread_scaling_1x1s": scaling, "thread_pick_2x4s_seconds": {str(k): v for k, v in pick.items()},
        "gpu_vs_oracle_maxabs": gpu_err(ci8, ro8),
        "config0_1x1s": {"value": 320 * 50 / m1, "unit": "samples/s", "rtf": m1 / 1.0,
                         "sample": f"1 utterance x 1 s, median of 3 runs ({m1:.3f} s)",
                         "gpu_vs_oracle_maxabs": gpu_err(ci1, ro1)},
        "host_cpus": host_cpus, "cpu_model": _cpu_model_string(),
        "threads_note": "headline value at the torch intra-op thread count that was fastest on the 1 x 1 s sample of "
                        "configs[0] (`thread_scaling_1x1s`: 8 ... 128 threads; compare it with config0_1x1s, not with the "
                        "headline sample)",
    }


# ------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_children(argv, args.gpus)

    # stdout carries exactly ONE line: RCCL prints a version banner with C-level printf when a communicator comes up,
    # so file descriptor 1 is pointed at stderr for the whole run and the result line is written to the saved
    # descriptor at the end
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)
    result, wl = run_bench(args, VocoderWorkload)
    import torch.distributed as dist
    if result is not None:
        _progress(f"timed region done: {result['ms_per_step']:.2f} ms per step")
        if not args.no_roofline:
            vocoder_roofline(args, wl, result)
            _progress(f"roofline pass done: frac {result['roofline']['frac']:.3f}")
        if result["n_gpus"] == 1 and not args.no_cpu_baseline:
            cpu_baseline(args, wl, result)
        if result["n_gpus"] == 1 and not args.no_extra:
            from tools import bench_extra
            extra = {}
            _progress("extra_configs: vocoder 1 x 1 s")
            extra["vocoder_b1_1s"] = bench_extra.vocoder_b1_1s(wl.dev, steps=20, net=wl.model)
            _progress("extra_configs: vocoder 1 x 4 s")
            extra["vocoder_b1_4s"] = bench_extra.vocoder_b1(wl.dev, seconds=4.0, steps=20, net=wl.model)
            _progress("extra_configs: vocoder + SpeechSR48, batch 32")
            extra["sr48_b32"] = bench_extra.sr48_b32(wl.dev, steps=5, net=wl.model)
            del wl
            _progress("extra_configs: full TTS, batch 16")
            from megatts2_hierspeechpp_amd import inference_plm as IP, synth
            import torch
            models = IP.TtsModels(bench_extra.VOC_CFG, bench_extra.TTV_CFG)
            models.load_state_dict({k: torch.from_numpy(synth.synth_tensor(k, tuple(v.shape), 0))
                                    for k, v in models.state_dict().items()})
            models.finalize(torch_device())
            extra["tts_b16"] = bench_extra.tts_b16(torch_device(), steps=3, models=models)
            _progress("extra_configs: full TTS, two batches of 16 in flight")
            extra["tts_2x16"] = bench_extra.tts_2x16(torch_device(), steps=3, models=models)
            _progress("extra_configs: full TTS, batch 1")
            extra["tts_b1"] = bench_extra.tts_b1(torch_device(), steps=3, models=models)
            _progress("extra_configs: TTS from the prompt waveform with the denoiser (SURVEY 8f N4)")
            extra["tts_prompt_denoise"] = bench_extra.tts_prompt_denoise(torch_device(), steps=3, models=models)
            del models
            _progress("extra_configs: voice conversion 1 x 4 s (SURVEY 8f N2)")
            extra["vc_b1_4s"] = bench_extra.vc_b1_4s(torch_device(), steps=10)
            result["extra_configs"] = extra
        sys.stdout.flush()
        os.write(out_fd, (json.dumps(result) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


def torch_device():
    import torch
    return torch.device("cuda", torch.cuda.current_device())


if __name__ == "__main__":
    sys.exit(main())
