#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rb; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "readout16 or merge_vproj" > $O/ops.log 2>&1; tail -5 $O/ops.log
python tools/kbench.py r16 2>&1 | tail -6
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/rb/bench.json"))
for k in ('ms_per_step','ms_per_step_batches','ms_per_step_pipelined','ms_per_step_same_buffers','ms_per_step_plan_miss','whole_step_hbm_frac','parity'):
    print(k, d.get(k))
print(d['roofline']['mean_launch_ms'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 200 --warmup 20 --no-extras > $O/trace.log 2>&1
cd $R
python3 tools/timeline_all.py $(ls $O/trace/*/*kernel_trace.csv | head -1) 9
