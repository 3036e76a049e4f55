#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-shard_trace}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
HICOM_BENCH_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench.log 2>&1
python3 tools/shard_trace.py $O/tr > $O/trace.txt; cat $O/trace.txt
