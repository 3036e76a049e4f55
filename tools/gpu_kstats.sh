#!/bin/bash
# Dev: per-kernel durations of the bench loop under environment settings.  usage: bash tools/gpu_kstats.sh <tag> "ENV=.. ENV=.." ["ENV=.."...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
i=0
for envs in "$@"; do
  i=$((i+1))
  export $envs
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$i -- python3 bench.py --steps 500 --warmup 50 --no-extras > $O/stats_$i.log 2>&1
  for v in $envs; do unset ${v%%=*}; done
  f=$(ls $O/stats_$i/*/*kernel_stats.csv | head -1)
  cp $f $O/kernel_stats_$i.csv
  echo "== $envs"
  python3 - $f <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hicom::" in r["Name"]]
for r in rows[:4]:
    print(f"  {r['Name'][:60]:60s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:7.2f}")
PY
  python3 tools/step_trace.py $O/stats_$i
  grep '^{' $O/stats_$i.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  step %.2f us (under the profiler)' % (d['ms_per_step']*1e3))"
done
