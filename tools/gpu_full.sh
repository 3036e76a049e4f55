#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log | cut -c1-200
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/full/bench.json"))
for k in ('value','ms_per_step','ms_per_step_batches','ms_per_step_pipelined','ms_per_step_same_buffers','ms_per_step_plan_miss','whole_step_hbm_frac','parity','secondary','neighbours','cpu_baseline'):
    print(k, d.get(k))
print(d['roofline'])
PY
HICOM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-secondary --no-cpu-baseline 2>$O/dist.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dist world1:', d['ms_per_step'], d.get('ms_per_step_pipelined'))"
