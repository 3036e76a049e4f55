#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rf; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "fused_stream" > $O/ops.log 2>&1; tail -4 $O/ops.log | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/par.log 2>&1; tail -4 $O/par.log | cut -c1-300
python tools/kbench.py fused 2>&1 | tail -2
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/rf/bench.json"))
for k in ('ms_per_step','ms_per_step_batches','ms_per_step_same_buffers','whole_step_hbm_frac','parity'):
    print(k, d.get(k))
print(d['roofline'])
PY
