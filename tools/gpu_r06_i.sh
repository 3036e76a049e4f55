#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_i; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_backward.py -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^FAILED" $O/pytest.log | cut -c1-250 | tail -12
grep -E "^E  " $O/pytest.log | head -12 | cut -c1-250
(timeout 420 python3 tools/flake_loop.py 9 > $O/flake3.log 2>&1; tail -3 $O/flake3.log)
