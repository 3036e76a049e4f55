#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rc; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "readout16 or merge_vproj" > $O/ops.log 2>&1; tail -4 $O/ops.log
echo "ring 8:"; python tools/kbench.py r16 2>&1 | tail -5

timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/rc/bench.json"))
for k in ('ms_per_step','ms_per_step_batches','ms_per_step_same_buffers','ms_per_step_plan_miss','whole_step_hbm_frac','parity'):
    print(k, d.get(k))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
