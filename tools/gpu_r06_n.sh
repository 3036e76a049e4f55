#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_n; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_surface.py -q -k "chain or baseline_configs or c4 or hidden_width or bit_stable or bit_identical or three_and_four" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^FAILED" $O/pytest.log | cut -c1-220 | tail -8; grep -E "^E  " $O/pytest.log | head -8 | cut -c1-250
python3 tools/c4_step.py 2000 2>/dev/null | tail -1
HICOM_CHAIN_WIDE=0 python3 tools/c4_step.py 2000 2>/dev/null | tail -1
python3 tools/c4_step.py 1000 64 2>/dev/null | tail -1
HICOM_CHAIN_WIDE=0 python3 tools/c4_step.py 1000 64 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 tools/c4_step.py 500 > $O/c4.log 2>&1
head -8 $(ls $O/c4/*/*kernel_stats.csv | head -1) | cut -c1-170
