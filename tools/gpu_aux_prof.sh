#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ax; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/kbench.py aux > $O/log 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/ax/t/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "readout16" in r["Kernel_Name"] or "linear_rows" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
# groups of 55 launches (5 warm + 50)
for g in range(0,len(d),55):
    seg=d[g:g+55]
    print(rows[g]["Kernel_Name"][:40], len(seg), "median %.2f us  min %.2f" % (sorted(seg)[len(seg)//2], min(seg)))
PY
