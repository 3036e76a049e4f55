#!/bin/bash
# Dev: rocprofv3 kernel stats of an arbitrary python command.  usage: bash tools/gpu_prof_cmd.sh <tag> <python args...>
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 "$@" > $O/run.log 2>&1
tail -3 $O/run.log
f=$(ls $O/stats/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f"  {r['Name'][:86]:86s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
