"""Dev tool: total time of K back-to-back drop-in steps between two device synchronisations, K = 1 .. 160 (the fixed cost of a timed
region: ~31 us of launch ramp + synchronisation at the benchmark shape, i.e. +1.6 us per step at the driver's K = 20)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
cfg = bench.release_config(896, 64)
m = bench.make_projector(cfg, dev)
sets = [(torch.randn(64, 27, 27, 1152, device=dev).bfloat16(), torch.randn(64, 27, 27, 1152, device=dev).bfloat16(), torch.randn(1152, device=dev).bfloat16()) for _ in range(3)]
i = [0]
def step():
    a, b, g = sets[i[0] % 3]; i[0] += 1
    return m(a, b, g, "video", None)
with torch.no_grad():
    for _ in range(400): step()
    for K in (1, 2, 5, 10, 20, 40, 80, 160):
        ts = []
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K): step()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        ts.sort()
        print(f"K={K:4d}: total {ts[4]:8.1f} us  per step {ts[4] / K:7.2f}")
    # host cost of a step (no sync)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"host enqueue per step {(t1 - t0) / 200 * 1e6:.1f} us")
