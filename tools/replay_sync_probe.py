import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
args = bench.parse_args([])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
wl = bench.VocoderWorkload(args, 0, 1, dev); wl.model.finalize(dev); wl.prepare(0, args.batch)
step = wl.make_step()
for _ in range(3): step()
torch.cuda.synchronize()
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); a = 1e3 * (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        step(); torch.cuda.synchronize()
    b = 1e3 * (time.perf_counter() - t0) / 10
    ev = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize(); ev.append(e0.elapsed_time(e1))
    print(f"back-to-back {a:.2f} | sync after each {b:.2f} | event per synced replay: median {sorted(ev)[5]:.2f}")
