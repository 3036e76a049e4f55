#!/bin/bash
# Dev: per-kernel durations (rocprofv3 kernel trace) and wall time of the non-release recipes at C2.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/modes; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in ${1:-off coarse fine adaptkv}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -- python3 $R/tools/modes_one.py $mode > $O/$mode.log 2>&1
  python3 - $O/$mode $mode <<'PY2'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
print("==", sys.argv[2])
for r in csv.DictReader(open(f)):
    if "hicom::" in r["Name"]:
        print(f"  {r['Name'][:64]:64s} calls/fwd {int(r['Calls'])/12:5.1f}  avg {float(r['AverageNs'])/1e3:7.1f} us  per fwd {int(r['TotalDurationNs'])/12e3:7.1f}")
PY2
done
python3 - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
dev = torch.device("cuda", 0)
for name, (ptype, guide) in {"off": ("local43_global32", None), "coarse": ("local43_global32", "coarse"), "fine": ("local43_global32", "fine"), "adaptkv": ("local43_adaptkv_global32", "direct")}.items():
    ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff)
    g = torch.randn(64, 1152, device=dev).bfloat16() if guide == "fine" else torch.randn(1152, device=dev).bfloat16()
    cfg = bench.release_config(896, 64); cfg.mm_projector_type = ptype; cfg.use_guide = guide
    m = bench.make_projector(cfg, dev)
    with torch.no_grad():
        for _ in range(5): m(ff, fe, g, "video", None)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): m(ff, fe, g, "video", None)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
    print(f"{name}: {best * 1e6:.0f} us")
PY
