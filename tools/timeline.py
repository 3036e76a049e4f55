"""Prints one steady-state step of a rocprofv3 kernel trace as a timeline (dev tool)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hicom" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[3] if len(sys.argv) > 3 else "linear_rows"
idx = [i for i, r in enumerate(rows) if "fused_ring" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2] - 2     # the two query-prep launches precede the stream kernel
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + int(sys.argv[2]) if len(sys.argv) > 2 else i0 + 16]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f q=%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Kernel_Name"][:70]))
