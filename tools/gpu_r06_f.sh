#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_f; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_gpu_surface.py tests/test_gpu_parity.py -q -k "marginals_form or handoff or shard or c3_c5 or sharded or world1 or eight_rank or bench_distributed or race" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^FAILED" $O/pytest.log | cut -c1-220 | tail -8
for i in 1 2; do HICOM_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>$O/dist.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('world-1 distributed branch: headline (pipelined) %.2f us  median %.2f  joined %.2f  world_size %s  handoff %s' % (d['ms_per_step']*1e3, d['ms_per_step_batches']['median']*1e3, d['ms_per_step_joined']['median']*1e3, d['world_size'], d['handoff_failures']))"; done
HICOM_SHARD_DIRECT_AG=0 HICOM_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>$O/dist2.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  (c10d all-gather) pipelined %.2f us  median %.2f  joined %.2f' % (d['ms_per_step']*1e3, d['ms_per_step_batches']['median']*1e3, d['ms_per_step_joined']['median']*1e3))"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain no-extras: %.2f us' % (d['ms_per_step']*1e3))"
python3 tools/shard_host.py 2>&1 | grep -v amdgpu | head -4
bash tools/gpu_shard_trace.sh r06_f_tr 2>&1 | head -50
