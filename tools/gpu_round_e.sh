#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/re; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_head.py -x -q -s > $O/head.log 2>&1; grep -n "passed\|failed\|\[head\]\|^E " $O/head.log | head -20
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "adapt or G5" > $O/adapt.log 2>&1; tail -3 $O/adapt.log
python tools/modes.py 2>&1 | tail -6
