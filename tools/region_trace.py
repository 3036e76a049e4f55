"""Dev tool: per-step timeline of the LAST K release steps in front of the roofline launches of a `bench.py --steps K --no-extras` run under
`rocprofv3 --kernel-trace` (the timed region): per-step kernel durations, gaps between the launches, start-to-start period."""
import csv, glob, sys
d, K = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
pat = ("query_prep", "fused_ring", "merge_vproj", "readout16", "readout16")
steps = []
for k in range(len(rows) - 4):
    if all(p in names[k + j] for j, p in enumerate(pat)):
        st = [int(rows[k + j]["Start_Timestamp"]) for j in range(5)]
        en = [int(rows[k + j]["End_Timestamp"]) for j in range(5)]
        steps.append((st, en))
print(f"{len(steps)} release steps in the trace")
# the timed region = the last K steps of the step loop (the roofline launches behind it are bare ring launches: no query_prep in front)
reg = steps[-K:]
pre = steps[-K - 40:-K]
def show(tag, ss):
    per = [(ss[i + 1][0][0] - ss[i][0][0]) / 1e3 for i in range(len(ss) - 1)]
    dur = [[(e - s) / 1e3 for s, e in zip(*x)] for x in ss]
    gaps = [[(x[0][j + 1] - x[1][j]) / 1e3 for j in range(4)] for x in ss]
    print(tag, "period us:", " ".join(f"{p:.1f}" for p in per))
    print(tag, "kernel sums:", " ".join(f"{sum(d_):.1f}" for d_ in dur))
    print(tag, "gap sums   :", " ".join(f"{sum(g_):.1f}" for g_ in gaps))
    import statistics as S
    print(tag, "median kernels:", [round(S.median(c), 2) for c in zip(*dur)], "median gaps:", [round(S.median(c), 2) for c in zip(*gaps)])
    print(tag, "span first start -> last end: %.1f us = %.2f per step" % ((ss[-1][1][4] - ss[0][0][0]) / 1e3, (ss[-1][1][4] - ss[0][0][0]) / 1e3 / len(ss)))
show("pre   ", pre)
show("region", reg)
