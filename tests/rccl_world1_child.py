"""Child process of test_rccl_world1_on_device (tests/test_gpu_parity.py): ONE rank that forms a real RCCL process
group on cuda:0 and drives every collective the 8-GPU job of BASELINE.json configs[4] depends on -- code that a
world-size-1 run otherwise skips (the reference's own "multi-GPU" is four pinned copies of a script,
inference_plm.py:336-339; its only NCCL init is train_ms.py:106).

    python tests/rccl_world1_child.py            (started BEFORE anything in this process touched the GPU)

Sequence: init_process_group("nccl", device_id=cuda:0) -> finalize_distributed through its broadcast branch (chunked
dist.broadcast on views of the packed weight arena) -> barrier_max / gather_floats on device tensors -> the golden case
`infer_config1` eagerly -> the same step captured into a hipGraph AFTER the collectives ran and replayed twice ->
a second broadcast + all_reduce after the replay (the communicator survives a capture) -> one JSON line."""
import json
import os
import socket
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import helpers as H
    from megatts2_hierspeechpp_amd import parallel

    rank, local_rank, world = parallel.init_distributed("nccl", force=True)
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    assert parallel.collectives_on()
    dev = torch.device("cuda", local_rank)

    meta, arrays = H.load_fixture("infer_config1")
    mod = H.build_module(meta)
    mod.load_state_dict(H.synth_sd(meta), strict=True)
    tm = {}
    arena = parallel.finalize_distributed(mod, dev, src=0, timings=tm, force_collective=True)
    assert tm["broadcast_ms"] is not None and tm["broadcast_ms"] > 0, tm
    before = arena.buffer.clone()
    # several chunks + a ragged tail over views of the device arena, as the N > 1 path issues them
    parallel.broadcast_buffer(arena.buffer, 0, chunk_elems=(1 << 20) + 3)
    torch.cuda.synchronize(dev)
    assert torch.equal(before, arena.buffer)
    assert parallel.barrier_max(1.5, dev) == 1.5
    assert parallel.gather_floats(2.5, dev) == [2.5]
    parallel.barrier()

    d = lambda k: torch.from_numpy(arrays[k]).to(dev)
    ins = (d("mel"), d("w2v"), d("lengths"), d("f0"))
    noise = d("noise")
    refs = H.outputs(arrays)

    def step():
        return mod.infer(*ins, noise=noise)

    def err(outs):
        return [float(np.abs(o.detach().cpu().numpy() - r).max()) for o, r in zip(outs, refs)]

    with torch.no_grad():
        eager = err(step())
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static = step()
        graph.replay()
        graph.replay()
        torch.cuda.synchronize(dev)
        replay = err(static)
    # the communicator still works after a capture + replay on the same device
    parallel.broadcast_buffer(arena.buffer, 0)
    after = parallel.barrier_max(3.25, dev)
    torch.cuda.synchronize(dev)
    assert after == 3.25 and torch.equal(before, arena.buffer)
    tols = [H.tol_for(r) for r in refs]
    print(json.dumps({"rccl_world": dist.get_world_size(), "backend": dist.get_backend(),
                      "broadcast_ms": tm["broadcast_ms"], "pack_ms": tm["pack_ms"], "arena_mb": arena.buffer.numel() * 4 / 1e6,
                      "eager_err": eager, "replay_err": replay, "tol": tols}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    ok = all(e <= t for e, t in zip(eager, tols)) and all(e <= t for e, t in zip(replay, tols))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
