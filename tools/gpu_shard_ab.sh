#!/bin/bash
# Dev A/B: frame-sharded step at world size 1 through RCCL, the STREAM -> comm event without / with the system-scope fence
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-shard_ab}; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests -m gpu -q -x -k "shard or dist or all_gather" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
one() { HICOM_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 1000 --warmup 100 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: pipelined %.2f us  joined %.2f' % (d['ms_per_step']*1e3, d['ms_per_step_joined']['median']*1e3))"; }
for i in 1 2; do
  HICOM_SHARD_EVENT_NOFENCE=0 one "torch event (system fence)"
  HICOM_SHARD_EVENT_NOFENCE=1 one "device-scope event        "
done | tee $O/shard_ab.txt
