"""Dev tool: per-kernel medians of the release step from a rocprofv3 kernel trace -- five launches (query_prep | ring | merge | GEMM 1 | GEMM 2,
round 4), four (query_prep | ring | GEMM 1 + merge role | GEMM 2 + chain role) or three (query_prep | ring | fused tail; round 5)."""
import csv, glob, statistics as st, sys
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    for pat, label in ((("query_prep", "fused_ring", "merge_vproj", "readout16", "readout16"), "prep %.2f  ring %.2f  merge %.2f  gemm1 %.2f  gemm2 %.2f"),
                       (("query_prep", "fused_ring", "readout16", "readout16"), "prep %.2f  ring %.2f  gemm1+merge %.2f  gemm2+chain %.2f"),
                       (("query_prep", "fused_ring", "readout_tail"), "prep %.2f  ring %.2f  tail %.2f")):
        n = len(pat)
        seq = []
        for k in range(len(rows) - n + 1):
            if all(p in names[k + j] for j, p in enumerate(pat)):
                dd = [int(rows[k + j]["End_Timestamp"]) - int(rows[k + j]["Start_Timestamp"]) for j in range(n)]
                seq.append(dd + [int(rows[k + n - 1]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])])
        if len(seq) < 10:
            continue
        seq = seq[len(seq) // 2:]
        med = [st.median(c) / 1e3 for c in zip(*seq)]
        print(("  trace: " + label + " | first start -> last end %.2f us  (%d steps)") % (*med, len(seq)))
