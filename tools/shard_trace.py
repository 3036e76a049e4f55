"""Dev: kernel timeline of the frame-sharded step at world size 1 (bench.py's distributed branch under rocprofv3 --kernel-trace):
prints ~3 consecutive steps of the PIPELINED loop and of the JOINED loop (start offset, duration, queue, kernel), and per-kernel medians.
usage: python3 tools/shard_trace.py <rocprof output dir>"""
import csv, glob, statistics as st, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ", "").replace("hicom::", "").split("(")[0][:46]
by = {}
for r in rows:
    by.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("per-kernel medians (us), calls:")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {k:48s} {st.median(v):8.2f}  x{len(v)}")
rings = [i for i, r in enumerate(rows) if "fused_ring" in r["Kernel_Name"]]
def show(lo_ring, n, title):
    i0 = rings[lo_ring] - 1
    i1 = rings[lo_ring + n]
    t0 = int(rows[i0]["Start_Timestamp"])
    print(title)
    for r in rows[i0:i1]:
        print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e3:8.2f}  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.2f} us  q{r.get('Queue_Id', '?'):>3s}  {short(r['Kernel_Name'])}")
# the bench runs: pre-warm-up + timed region of the pipelined loop first, then batches, then the joined loop (3 x per steps)
show(len(rings) // 4, 3, "pipelined loop (three steps):")
gaps = [int(rows[rings[k + 1]]["Start_Timestamp"]) - int(rows[rings[k]]["Start_Timestamp"]) for k in range(len(rings) - 1)]
print("ring-to-ring start distance: median %.2f us over the first half, %.2f over the last tenth" % (st.median(gaps[:len(gaps) // 2]) / 1e3, st.median(gaps[-len(gaps) // 10:]) / 1e3))
show(len(rings) - 8, 3, "late in the run (three steps):")
