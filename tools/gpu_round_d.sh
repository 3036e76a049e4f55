#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rd; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -x -q -k "readout16 or merge_vproj or sharded_forward or c2_full or golden" > $O/ops.log 2>&1; tail -3 $O/ops.log
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/rd/bench.json"))
for k in ('ms_per_step','ms_per_step_batches','ms_per_step_same_buffers','whole_step_hbm_frac'):
    print(k, d.get(k))
PY
bash tools/gpu_trace.sh
