#!/bin/bash
# Dev (round 4): the whole GPU test suite + the full default bench line (secondary recipes, neighbours, CPU baseline).
TAG=${1:-r04_full}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log | cut -c1-220
timeout 1200 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
for k in ('value','ms_per_step','ms_per_step_batches','ms_per_step_pipelined','ms_per_step_same_buffers','ms_per_step_plan_miss','whole_step_hbm_frac','whole_step_hbm_frac_median','parity','cpu_baseline'):
    print(k, d.get(k))
print(d['roofline'])
for k, v in d.get('secondary', {}).items(): print(k, v)
for k, v in d.get('neighbours', {}).items(): print(k, v)
PY
