#!/bin/bash
# Dev: per-kernel table of one recipe's training steps (captured hipGraph backward by default).  usage: bash tools/gpu_train_kernels.sh <recipe> [steps]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trk_${1:-release}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/train_kernels.py ${1:-release} ${2:-20} > $O/run.log 2>&1
tail -1 $O/run.log
python3 - $(ls $O/*/*kernel_stats.csv | head -1) ${2:-20} <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); steps = int(sys.argv[2]) + 3
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("device time per step %.1f us over %d kernels/step" % (tot / steps / 1e3, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:22]:
    print("  %-70s calls/step %5.1f  avg %7.2f us  per step %7.1f us" % (r["Name"][:70], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e3))
PY
