"""Dev tool (round 5, verdict item 1): what does the first timed region of a fresh process pay?

Fresh process, the bench's own module / input sets / step().  Prints one JSON object with
  stamps_first   per-step durations (us) of the first 400 steps of the process from hipEvent stamps (one event per step)
  windows        per-step time (us) of 150 back-to-back `timed(step, 20)` regions (sync | 20 steps | sync) with the wall time since the first launch
  after_idle     the same region after the GPU sat idle for 5 / 50 / 500 ms
  after_queue    the round-4 protocol: 305 un-fenced steps, then the timed region
  k_sweep        K = 20 / 50 / 200 interleaved (fixed cost of a region)
  stamps_late    per-step hipEvent stamps of a later 50-step batch
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = bench.release_config(896, 64)
m = bench.make_projector(cfg, dev)
gen = torch.Generator(device=dev).manual_seed(1234)
sets = []
for i in range(3):
    ff = torch.randn(64, 27, 27, 1152, device=dev, generator=gen).to(torch.bfloat16)
    fe = torch.randn(64, 27, 27, 1152, device=dev, generator=gen).to(torch.bfloat16)
    g = torch.randn(1152, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + i)).to(torch.bfloat16)
    sets.append((ff, fe, g))
cnt = [0]


def step():
    a, b, g = sets[cnt[0] % 3]
    cnt[0] += 1
    return m(a, b, g, "video", None)


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def stamps(n):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    s = torch.cuda.current_stream()
    evs[0].record(s)
    for i in range(n):
        step()
        evs[i + 1].record(s)
    torch.cuda.synchronize()
    return [round(evs[i].elapsed_time(evs[i + 1]) * 1e3, 2) for i in range(n)]


import gc
res = {}
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
with torch.no_grad():
    gc.collect()
    gc.disable()
    step()                      # plan build, table builds, lazy code-object loads
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    if mode in ("all", "stamps"):
        res["stamps_first"] = stamps(400)
    w = []
    for _ in range(150):
        us = timed(20)
        w.append([round((time.perf_counter() - t_first) * 1e3, 2), round(us, 2)])
    res["windows"] = w
    idle = {}
    for ms in (5, 50, 500):
        r = []
        for _ in range(3):
            torch.cuda.synchronize()
            time.sleep(ms * 1e-3)
            r.append(round(timed(20), 2))
            r.append(round(timed(20), 2))
        idle[str(ms)] = r
    res["after_idle"] = idle
    q = []
    for _ in range(8):
        for _ in range(305):
            step()
        q.append(round(timed(20), 2))
    res["after_queue"] = q
    ks = {20: [], 50: [], 200: []}
    for _ in range(9):
        for K in ks:
            ks[K].append(round(timed(K), 2))
    res["k_sweep"] = {str(k): sorted(v) for k, v in ks.items()}
    res["stamps_late"] = stamps(50)
    res["final_windows"] = [round(timed(20), 2) for _ in range(10)]
print(json.dumps(res))
