#!/bin/bash
# Dev A/B: stream-to-stream events without / with the system-scope fence (HICOM_EVENT_NOFENCE), secondary recipes + sharded step + training step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-events}; mkdir -p $O; cd $R
for i in 1 2; do for v in 0 1; do
  HICOM_EVENT_NOFENCE=$v python3 bench.py --gpus 1 --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NOFENCE=$v step %.2f us' % (d['ms_per_step']*1e3), ' secondary', {k[:24]: (round(v['ms_per_forward'], 4), round(v.get('train_step_ms', 0), 3)) for k, v in d['secondary'].items()}, ' train', round(d['neighbours']['train_step']['params_only_ms'],4), ' deferred', d.get('ms_per_step_deferred'))"
done; done | tee $O/events_ab.txt
