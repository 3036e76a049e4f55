#!/bin/bash
# round 6: the closing run -- full GPU suite, smoke, the driver's command, the world-1 distributed branch, the evidence bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final; rm -rf $O; mkdir -p $O; cd $R
timeout 1200 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=" $O/pytest.log | tail -3 | cut -c1-300; grep -E "^FAILED" $O/pytest.log | cut -c1-200 | head -20
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
bash tools/gpu_driver_cmd.sh r06_final_drv 2>&1 | tail -12
cp gpurun_out/r06_final_drv/full.json gpurun_out/r06_d_driver_cmd.json
HICOM_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r06_d_dist_world1.json
