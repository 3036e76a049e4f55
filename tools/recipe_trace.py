"""Dev tool: the kernel timeline of ONE forward of a secondary recipe (coarse | fine | off | adaptkv | image) from a rocprofv3 kernel trace.
On the GPU box:   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/recipe_trace.py run fine
then              python3 tools/recipe_trace.py report $O"""
import csv, glob, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch, bench, time
    dev = torch.device("cuda", 0)
    ug = sys.argv[2]
    cfg = bench.release_config(896, 64)
    cfg.mm_projector_type = "local43_adaptkv_global32" if ug == "adaptkv" else "local43_global32"
    cfg.use_guide = {"off": None, "adaptkv": "direct", "image": "direct"}.get(ug, ug)
    m = bench.make_projector(cfg, dev)
    T, modal = (1, "image") if ug == "image" else (64, "video")          # "image": the release recipe on ONE image (windows of 1 x 3 x 3)
    ff = torch.randn(T, 27, 27, 1152, device=dev).bfloat16()
    fe = torch.randn(T, 27, 27, 1152, device=dev).bfloat16()
    g = torch.randn(64, 1152, device=dev).bfloat16() if ug == "fine" else torch.randn(1152, device=dev).bfloat16()
    with torch.no_grad():
        for _ in range(20):
            m(ff, fe, g, modal, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            m(ff, fe, g, modal, None)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("host enqueue %.1f us / forward, wall %.1f us / forward" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
        torch.cuda.synchronize()
        time.sleep(0.01)
        m(ff, fe, g, modal, None)          # ONE isolated forward: the last kernels of the trace
        torch.cuda.synchronize()
else:
    f = glob.glob(sys.argv[2] + "/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last forward: walk back from the end until a gap > 2 ms
    last = [rows[-1]]
    for r in reversed(rows[:-1]):
        if int(last[-1]["Start_Timestamp"]) - int(r["End_Timestamp"]) > 2_000_000:
            break
        last.append(r)
    last.reverse()
    t0 = int(last[0]["Start_Timestamp"])
    for r in last:
        print("  %8.1f -> %8.1f us  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"),
                                                r["Kernel_Name"][:90]))
