import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
cfg = bench.release_config(896, 64)
m = bench.make_projector(cfg, dev)
g = torch.randn(1152, device=dev).bfloat16()
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff)
vids = [(ff, fe), (fe, ff), (ff.flip(0).contiguous(), fe), (ff, fe)]
with torch.no_grad():
    want = [m(a, b, g, "video", None).clone() for a, b in vids]
    for rep in range(3):
        for lanes in (2, 3):
            hs = [m.forward_async(a, b, g, "video", None, lanes=lanes) for a, b in vids for _ in range(2)]
            got = [h.wait() for h in hs]
            torch.cuda.synchronize()
            d = [float((o.float() - want[k // 2].float()).abs().max()) for k, o in enumerate(got)]
            rows = [int(((o != want[k // 2]).any(dim=1)).sum()) for k, o in enumerate(got)]
            for k, o in enumerate(got):
                idx = torch.nonzero((o != want[k // 2]).any(dim=1)).flatten().tolist()
                if idx: print("   k", k, "rows", idx[:20], "ncols", int((o[idx[0]] != want[k // 2][idx[0]]).sum()))
            print("rep", rep, "lanes", lanes, "maxdiff", ["%.2e" % x for x in d], "rows differing", rows)
