#!/bin/bash
# Dev: ring kernel check after an edit: parity subset, in-kernel phase stamps, bench (no extras).
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -x -q -k "fused or ring or c2 or direct or golden or matches_oracle" 2>&1 | tail -2
python tools/fused_trace.py 2>&1 | grep -E "kernel \(events|wait at \[A\] \(all|softmax\+marg|completion|loop total|reads\+scores|wait at \[B\] \(all"
for i in 1 2; do python bench.py --steps 1000 --warmup 50 --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('step', round(d['ms_per_step']*1e3,2), 'us  ring', round(r['mean_launch_ms']*1e3,2), 'us frac', round(r['frac'],3))"; done
