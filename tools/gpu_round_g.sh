#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rg; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/rg/bench.json"))
for k in ('ms_per_step','ms_per_step_batches','ms_per_step_same_buffers','whole_step_hbm_frac'):
    print(k, d.get(k))
print(d['roofline']['mean_launch_ms'], d['roofline']['frac'])
PY
bash tools/gpu_trace.sh 2>&1 | tail -16
