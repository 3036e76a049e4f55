#!/bin/bash
# Dev A/B: leading kernel arguments preloaded into SGPRs (the product build) against the s_load prologue (HICOM_KERNARG_PRELOAD=0 build), same box, interleaved.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-kernarg}; mkdir -p $O; cd $R
HICOM_KERNARG_PRELOAD=0 python3 -c "
from hicom_amd import build_native as bn
bn.build(lib_path='/tmp/libhicom_nopre.so', extra_flags=('-DHICOM_NOPRELOAD_BUILD',), verbose=False)"
one() { python3 bench.py --gpus 1 --steps 1000 --warmup 200 --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: step %.2f us  ring %.2f (%.3f)' % (d['ms_per_step']*1e3, d['roofline']['mean_launch_ms']*1e3, d['roofline']['frac']))"; }
for i in 1 2 3; do
  HICOM_NATIVE_LIB=/tmp/libhicom_nopre.so one "s_load prologue"
  one "preloaded     "
done | tee $O/ab.txt
timeout 300 python tools/tail_trace.py 2 > $O/tail2_pre.txt 2>&1; grep -E "first stage|main loop|kernargs|exit" $O/tail2_pre.txt
