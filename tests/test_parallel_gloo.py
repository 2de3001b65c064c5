"""CPU, world_size 2 over gloo: the N > 1 path of bench.py.

1. the collective pieces: rank 0 owns the packed weight buffer, every other rank receives it by a chunked
   broadcast, utterances are sharded in contiguous blocks, the timing reduction takes the max over ranks;
2. bench.py's own control flow (`bench.run_bench`) with a CPU stand-in workload: finalize_distributed with
   materialize=False on rank 1, shard_range over the global batch, barrier-bracketed timing, rank 0's result line;
3. `python bench.py --gpus N` without a launcher starts its ranks as a child process and relays rank 0's line
   (checked with the launcher module replaced by a stub: there is no GPU in this container)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    every = parallel.gather_floats(10.0 * rank, torch.device("cpu"))
    out.put((rank, ok, (lo, hi), tmax, every))
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
    assert all(ok for _, ok, _, _, _ in res)
    assert [span for _, _, span, _, _ in res] == [(0, 17), (17, 33)]
    assert all(t == 2.0 for _, _, _, t, _ in res)
    assert all(e == [0.0, 10.0] for *_, e in res)


# ------------------------------------------------------------------ bench.py's N = 2 control flow on CPU
class _Arena:
    def __init__(self, buffer):
        self.buffer = buffer


class _StandInModel(torch.nn.Module):
    """Same finalize() contract as the product models (hip_layers.finalize): every rank lays out an identical
    arena, only `materialize=True` fills it."""
    N = (1 << 18) + 5

    def finalize(self, device, materialize=True):
        buf = torch.zeros(self.N, dtype=torch.float32, device=device)
        if materialize:
            buf.copy_(torch.arange(self.N, dtype=torch.float32) % 251.0)
        self.arena = _Arena(buf)
        return self.arena


class _StandInWorkload:
    samples_per_utterance = 1000

    def __init__(self, args, rank, world, dev):
        self.model, self.rank, self.dev = _StandInModel(), rank, dev
        self.shard = None

    def prepare(self, lo, hi):
        self.shard = (lo, hi)
        self.x = torch.arange(lo, hi, dtype=torch.float32)

    def make_step(self):
        def step():
            # uses the broadcast weights: a rank that did not receive them fails check()
            return (self.x.sum() + self.model.arena.buffer.sum(),)
        return step

    def check(self, out):
        want = torch.arange(self.shard[0], self.shard[1], dtype=torch.float32).sum() + \
            (torch.arange(_StandInModel.N, dtype=torch.float32) % 251.0).sum()
        assert torch.equal(out[0], want), (self.rank, out[0], want)

    def describe(self, world):
        return {"workload": "cpu stand-in", "global_batch": 7 * world}


def _bench_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "7"])
    result, wl = bench.run_bench(args, _StandInWorkload, backend="gloo", device=torch.device("cpu"))
    out.put((rank, result, wl.shard))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_control_flow_world2_cpu_stand_in():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, line, shard0), (r1, none, shard1) = res
    assert none is None and line is not None            # only rank 0 reports
    assert shard0 == (0, 7) and shard1 == (7, 14)       # contiguous shards of the global batch 7 x 2
    assert line["n_gpus"] == 2 and line["rccl_world"] == 2 and line["scaling"] == "weak"
    assert line["steps"] == 3 and line["warmup"] == 1 and line["config"]["shard_of_rank0"] == [0, 7]
    assert line["broadcast_ms"] > 0 and line["rank_ms_per_step"]["min"] <= line["rank_ms_per_step"]["max"]
    assert line["pack_ms"] > 0 and line["broadcast_gbs"] > 0       # the collective alone, apart from the packing
    # whole-job value: the units ALL ranks processed over the max-over-ranks time
    assert abs(line["value"] - 14 * 1000 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]
    json.dumps(line)


def _force_worker(port, out):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    from megatts2_hierspeechpp_amd import parallel
    args = bench.parse_args(["--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--batch", "5"])
    result, wl = bench.run_bench(args, _StandInWorkload, backend="gloo", device=torch.device("cpu"))
    out.put((result, wl.shard, parallel.collectives_on(), dist.is_initialized()))
    dist.destroy_process_group()


def test_force_dist_runs_the_collectives_at_world_size_1():
    """`--force-dist`: the one-rank job forms its process group and goes through the broadcast branch, the barriers and
    the timing reductions (the GPU twin of this test runs it over RCCL on the device)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_worker, args=(_free_port(), q))
    p.start()
    line, shard, coll, init = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0 and coll and init and shard == (0, 5)
    assert line["rccl_world"] == 1 and line["backend"] == "gloo" and line["broadcast_ms"] > 0 and line["broadcast_gbs"] > 0


def _bench_worker8(rank, world, port, out, global_batch):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    import bench
    argv = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "32"]
    if global_batch is not None:
        argv += ["--global-batch", str(global_batch)]
    args = bench.parse_args(argv)
    result, wl = bench.run_bench(args, _StandInWorkload, backend="gloo", device=torch.device("cpu"))
    out.put((rank, result, wl.shard))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [None, 250])
def test_bench_control_flow_world8_cpu_stand_in(global_batch):
    """BASELINE.json configs[4]'s shape -- 8 ranks, 256 utterances, 32 per rank -- through bench.run_bench on gloo
    (VERDICT r03 item 1b), and an uneven global batch of 250 (shards of 32 / 32 / 31 x 6): every rank's shard is
    contiguous, the shards tile the batch, `value` counts what ALL ranks processed."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker8, args=(r, world, port, q, global_batch)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    total = 256 if global_batch is None else global_batch
    shards = [sh for _, _, sh in res]
    assert shards[0][0] == 0 and shards[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(shards, shards[1:]))            # contiguous, no gap, no overlap
    sizes = [hi - lo for lo, hi in shards]
    assert max(sizes) - min(sizes) <= 1 and (sizes == [32] * 8 if global_batch is None else sizes == [32, 32] + [31] * 6)
    lines = [r for _, r, _ in res]
    assert lines[0] is not None and all(ln is None for ln in lines[1:])     # only rank 0 reports
    line = lines[0]
    # a fixed --global-batch is a strong-scaling run (the total work does not grow with the rank count); batch x gpus is weak
    assert line["n_gpus"] == 8 and line["rccl_world"] == 8 and line["backend"] == "gloo"
    assert line["scaling"] == ("weak" if global_batch is None else "strong")
    assert line["config"]["shard_of_rank0"] == [0, 32] and line["config"]["global_batch"] == total
    assert line["broadcast_ms"] > 0 and line["pack_ms"] > 0
    assert abs(line["value"] - total * 1000 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    json.dumps(line)


def test_bench_gpus_n_starts_its_own_ranks(tmp_path):
    """WORLD_SIZE unset and --gpus 2: the parent starts `python -m torch.distributed.run ... bench.py` as a CHILD
    (before importing torch itself) and relays rank 0's line.  Here the child's `torch.distributed.run` is a stub
    package first on PYTHONPATH that records its argv and prints a result line."""
    fake = tmp_path / "torch" / "distributed"
    fake.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (fake / "__init__.py").write_text("")
    (fake / "run.py").write_text(
        "import json, os, sys\n"
        "assert 'WORLD_SIZE' not in os.environ\n"
        "print('some launcher chatter')\n"
        "print(json.dumps({'metric': 'stub', 'argv': sys.argv[1:]}))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = str(tmp_path)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                      # exactly one JSON line on stdout
    argv = json.loads(lines[0])["argv"]
    assert "--nproc-per-node=2" in argv and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-4:] == ["--gpus", "2", "--steps", "4"] and argv[-5].endswith("bench.py")
    assert "some launcher chatter" in p.stderr            # everything else is relayed on stderr
