#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tb}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/tail_bench.py > $O/run.log 2>&1
tail -3 $O/run.log
python3 tools/tail_bench.py --report $O/trace
