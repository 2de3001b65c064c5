"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- rank 0 owns the packed weight
buffer, every other rank receives it by broadcast, utterances are sharded in contiguous
blocks, and the timing reduction takes the max over ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from megatts2_hierspeechpp_amd import parallel
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    # the arena stand-in: rank 0 holds the packed weights, the others an empty layout
    n = (3 << 20) + 17
    buf = torch.arange(n, dtype=torch.float32) if rank == 0 else torch.zeros(n)
    parallel.broadcast_buffer(buf, src=0, chunk_elems=1 << 20)     # several chunks + a ragged tail
    ok = bool((buf == torch.arange(n, dtype=torch.float32)).all())
    lo, hi = parallel.shard_range(33, rank, world)
    tmax = parallel.barrier_max(1.0 + rank, torch.device("cpu"))
    out.put((rank, ok, (lo, hi), tmax))
    dist.destroy_process_group()


def test_broadcast_shard_and_timing_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert [span for _, _, span, _ in res] == [(0, 17), (17, 33)]
    assert all(t == 2.0 for *_, t in res)
