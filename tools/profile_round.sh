#!/bin/bash
# Dev tool: the per-round evidence set -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/).
# usage (GPU box): bash tools/profile_round.sh r02_a
TAG=${1:-r02_x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 1000 --warmup 50 --no-extras > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc/fetch -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc/write -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc $R/gpurun_out/${TAG}_pmc_hbm_traffic.json 64
# MFMA utilisation: release recipe (ring kernel, planes GEMMs) + guide-off recipe (wide global stream kernel), own PMC passes
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma/direct -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/mfma1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma/off -- python3 tools/modes_one.py off > $O/mfma2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma/head -- python3 -m pytest tests/test_gpu_head.py -q -k benchmark > $O/mfma3.log 2>&1
python3 tools/pmc_mfma.py $O/mfma $R/gpurun_out/${TAG}_mfma_util.json
# BASELINE configs[3]'s compressor shape (32 frames, hidden 3584): kernel table + whole-step fraction
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 tools/c4_step.py 500 > $O/c4.log 2>&1
cp $(ls $O/c4/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_c4_kernel_stats.csv
python3 tools/c4_step.py 2000 2>/dev/null | tail -1 > $R/gpurun_out/${TAG}_c4_step.txt; cat $R/gpurun_out/${TAG}_c4_step.txt
python3 tools/step_trace.py $O/stats > $R/gpurun_out/${TAG}_step_trace.txt; cat $R/gpurun_out/${TAG}_step_trace.txt
timeout 900 python3 bench.py 2> $O/bench.err | grep '^{' > $R/gpurun_out/${TAG}_bench.json
cut -c1-600 $R/gpurun_out/${TAG}_bench.json
head -8 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
