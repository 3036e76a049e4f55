#!/bin/bash
# Dev tool: the per-round evidence set -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/).
# usage (GPU box): bash tools/profile_round.sh r01_f
TAG=${1:-r01_x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc/fetch -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc/write -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph > $O/write.log 2>&1
python3 tools/pmc_traffic.py $O/pmc $R/gpurun_out/${TAG}_pmc_hbm_traffic.json 64
timeout 900 python3 bench.py 2> $O/bench.err | grep '^{' > $R/gpurun_out/${TAG}_bench.json
cut -c1-400 $R/gpurun_out/${TAG}_bench.json
head -8 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
