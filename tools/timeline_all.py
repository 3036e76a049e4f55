"""Prints a window of a rocprofv3 kernel trace (all kernels, all queues) as a timeline (dev tool)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "fused_ring" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2] - 2
t0 = int(rows[i0]["Start_Timestamp"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f q=%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Kernel_Name"][:60]))
