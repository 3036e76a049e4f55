#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in small big; do
  export HICOM_DENSE_TILE=$v
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/$v -- python3 -m pytest $R/tests/test_gpu_head.py -x -q -k benchmark > $O/$v.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $O/$v dense16
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${v}f -- python3 -m pytest $R/tests/test_gpu_head.py -x -q -k benchmark > $O/${v}f.log 2>&1
  python3 $R/tools/pmc_summary.py $O/${v}f dense16
done
