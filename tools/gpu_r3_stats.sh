#!/bin/bash
# Dev tool (round 3): kernel stats of the bench loop + the GPU tests named on the command line.
# usage (GPU box): bash tools/gpu_r3_stats.sh <tag> [pytest -k expression]
TAG=${1:-r03_x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
if [ -n "$2" ]; then timeout 900 python3 -m pytest tests -m gpu -q -x -k "$2" 2>&1 | tail -15 > $O/tests.log; cat $O/tests.log | tail -5; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 1000 --warmup 50 --no-extras > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
head -12 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
timeout 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> $O/bench.err | grep '^{' > $R/gpurun_out/${TAG}_bench.json
python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/${TAG}_bench.json"))
print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_batches", "whole_step_hbm_frac_median")})
print(d["roofline"]["mean_launch_ms"], d["roofline"]["frac"])
PY
