#!/bin/bash
# Dev (round 4): full GPU test suite + a short bench line.  usage (GPU box): bash tools/gpu_r4_a.sh [tag] [pytest -k expr]
TAG=${1:-r04_a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
if [ -n "$2" ]; then
  timeout 2400 python3 -m pytest tests -m gpu -q -x -k "$2" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
else
  timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
fi
tail -8 $O/pytest.log | cut -c1-220
timeout 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> $O/bench.err | grep '^{' > $O/bench.json
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_batches", "ms_per_step_same_buffers", "whole_step_hbm_frac_median", "parity")})
print(d["roofline"]["mean_launch_ms"], d["roofline"]["frac"])
PY
