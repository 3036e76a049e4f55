#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 -m pytest $R/tests/test_gpu_head.py -x -q -k benchmark > $O/st.log 2>&1
head -6 $(ls $O/st/*/*kernel_stats.csv | head -1) | cut -c1-130
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc -- python3 -m pytest $R/tests/test_gpu_head.py -x -q -k benchmark > $O/pmc.log 2>&1
python3 $R/tools/pmc_summary.py $O/pmc dense16
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/pmc2 -- python3 -m pytest $R/tests/test_gpu_head.py -x -q -k benchmark > $O/pmc2.log 2>&1
python3 $R/tools/pmc_summary.py $O/pmc2 dense16
