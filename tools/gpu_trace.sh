#!/bin/bash
# in-situ kernel timeline of the joined forward on rotating inputs (rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 300 --warmup 20 --no-extras > $O/trace.log 2>&1
cd $R
python3 tools/timeline_all.py $(ls $O/trace/*/*kernel_trace.csv | head -1) 9
head -12 $(ls $O/trace/*/*kernel_stats.csv | head -1) | cut -c1-110
