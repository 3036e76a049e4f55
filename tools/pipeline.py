"""Dev tool: step time of the release recipe at the C2 shape, sync vs pipelined lanes, eager vs graph."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(1152, device=dev).bfloat16()
cfg = bench.release_config(896, 64)
m = bench.make_projector(cfg, dev)
n = 300
with torch.no_grad():
    ref = m(ff, fe, g, "video", None).clone()
    for graph in (False, True):
        m.graph_replay = graph
        for _ in range(5): out = m(ff, fe, g, "video", None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): out = m(ff, fe, g, "video", None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n): out = m(ff, fe, g, "video", None)
        host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        print("sync   graph=%d             %7.1f us/step (host enqueue %.1f us)  equal=%s" % (graph, dt * 1e6, host * 1e6, torch.equal(out, ref)))
        for lanes in (2, 3, 4):
            for _ in range(8): h = m.forward_async(ff, fe, g, "video", None, lanes=lanes)
            h.wait(); torch.cuda.synchronize(); t0 = time.perf_counter()
            hs = [m.forward_async(ff, fe, g, "video", None, lanes=lanes) for _ in range(n)]
            host = (time.perf_counter() - t0) / n
            outs = [h.wait() for h in hs]
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            ok = all(torch.equal(o, ref) for o in outs[-4:])
            print("async  graph=%d lanes=%d     %7.1f us/step (host enqueue %.1f us)  equal=%s" % (graph, lanes, dt * 1e6, host * 1e6, ok))
