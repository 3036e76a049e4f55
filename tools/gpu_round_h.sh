#!/bin/bash
# Dev: bench extras (train step) + world-1 sharded joined/pipelined numbers.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rh; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/rh/bench.json"))
print('step', d['ms_per_step'], 'pipelined', d.get('ms_per_step_pipelined'))
print(json.dumps(d.get('neighbours',{}).get('train_step')))
PY
HICOM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-secondary --no-cpu-baseline 2>$O/dist.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dist world1 headline:', d['ms_per_step'], 'joined', d.get('ms_per_step_joined'), 'pipelined', d.get('ms_per_step_pipelined'))"
