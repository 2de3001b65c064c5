"""Data-parallel batched synthesis: one process per GPU, utterances sharded in
contiguous blocks, weights shipped once by an RCCL broadcast of the packed arena.

The reference has no multi-GPU inference (four pinned copies of the script,
inference_plm.py:336-339; SURVEY.md §2 'Parallelism').  Utterances are independent, so
there is no data-path collective: the only communication is the start-up broadcast
(SURVEY.md §8e)."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


# When True, a world-size-1 job still forms its process group and runs every collective of the N > 1 path (the weight
# broadcast, the timing reductions): `bench.py --gpus 1 --force-dist` and the one-GPU RCCL test set it, so that the code
# the 8-GPU run depends on has executed on a device before the day a node is available.
FORCE_COLLECTIVES = False


def collectives_on() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def init_distributed(backend: str = "nccl", force: bool = False):
    """backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests.  ``force``: form the group at world size 1
    too (and keep every collective of the N > 1 path switched on, see FORCE_COLLECTIVES)."""
    global FORCE_COLLECTIVES
    rank, local_rank, world = env_world()
    if force:
        FORCE_COLLECTIVES = True
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of utterances owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_buffer(buf: torch.Tensor, src: int = 0, chunk_elems: int = 64 << 20, force: bool = False) -> torch.Tensor:
    """Broadcast one flat buffer in large chunks (xGMI is point-to-point: few, large
    transfers; 256 MiB chunks keep RCCL's ring pipelined without a huge staging need)."""
    if collectives_on() or (force and dist.is_initialized()):
        flat = buf.view(-1)
        for off in range(0, flat.numel(), chunk_elems):
            dist.broadcast(flat[off:off + chunk_elems], src=src)
    return buf


def barrier() -> None:
    if collectives_on():
        dist.barrier()


def finalize_distributed(model, device, src: int = 0, timings: dict = None, force_collective: bool = False,
                         derive: bool = True):
    """Rank ``src`` folds + packs the weights; every other rank only lays the arena out
    (identical offsets by construction) and receives the bytes by broadcast.

    ``timings`` (optional dict) receives ``pack_ms`` -- this rank's fold + pack + upload (or arena layout) -- and
    ``broadcast_ms`` -- the ``dist.broadcast`` loop alone, bracketed by a barrier and device synchronisation on both
    sides so that it starts when the slowest rank is ready and ends when the last byte has landed; ``None`` at
    world size 1 (no collective runs) unless ``force_collective`` / FORCE_COLLECTIVES asks for the broadcast branch on
    a one-rank group (the process group must exist: ``init_distributed(force=True)``).  ``derive``: every rank then
    derives the per-bin matrices of the frequency-domain convs from the taps it now holds (``derive_ms``,
    ``derived_mb``; ``broadcast_mb`` = the arena that travelled) instead of leaving that to the first call."""
    import time
    multi = collectives_on() or (force_collective and dist.is_initialized())
    on_gpu = torch.device(device).type == "cuda"
    sync = (lambda: torch.cuda.synchronize(device)) if on_gpu else (lambda: None)
    rank = dist.get_rank() if dist.is_initialized() else 0
    sync()
    t0 = time.perf_counter()
    res = model.finalize(device, materialize=(rank == src))
    sync()
    pack_ms = 1e3 * (time.perf_counter() - t0)
    # hip_layers.finalize leaves the arena on the model whatever the model's own finalize() returns (some return self)
    arena = getattr(model, "_hsp_arena", None)
    if arena is None:
        arena = res
    bc_ms = None
    if multi:
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        broadcast_buffer(arena.buffer, src, force=True)
        sync()
        dist.barrier()
        bc_ms = 1e3 * (time.perf_counter() - t0)
    # Derived data stays off the wire (the broadcast carries the folded weights, ~457 MB for the vocoder):
    # the per-bin matrices of the frequency-domain convs (3.6 GB at 42 convs) are computed from the broadcast taps by EVERY
    # rank with the same kernel (hsp_dftseg_weight_spectrum_f32), bit-identical across ranks.
    derive_ms = derived_mb = None
    if derive and on_gpu:
        from . import hip_layers
        t0 = time.perf_counter()
        with torch.cuda.device(device):
            n = hip_layers.prepare_fft(model)
        sync()
        derive_ms, derived_mb = 1e3 * (time.perf_counter() - t0), n * 4 / 1e6
    if timings is not None:
        timings["pack_ms"], timings["broadcast_ms"] = pack_ms, bc_ms
        timings["broadcast_mb"] = arena.buffer.numel() * 4 / 1e6 if getattr(arena, "buffer", None) is not None else None
        timings["derive_ms"], timings["derived_mb"] = derive_ms, derived_mb
    return arena


def barrier_max(value: float, device) -> float:
    """max over ranks of a python float (timing reduction of bench.py)."""
    if not collectives_on():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device) -> list:
    """every rank's python float, in rank order (per-rank spread of bench.py's step time)."""
    if not collectives_on():
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]
