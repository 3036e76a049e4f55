#!/bin/bash
# Dev (round 4): test subset + in-step kernel table + bench line.  usage: bash tools/gpu_r4_b.sh <tag> "<pytest -k expr>"
TAG=$1; KEXPR=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
if [ -n "$KEXPR" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -q -x -k "$KEXPR" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  tail -4 $O/pytest.log | cut -c1-200
fi
bash tools/gpu_kstats.sh $TAG/k "HICOM_NOP=1"
timeout 300 python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-secondary 2> $O/bench.err | grep '^{' > $O/bench.json
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("step %.2f us  median %.2f  ring %.2f us (%.3f)  parity %.2e" % (d["ms_per_step"] * 1e3, d["ms_per_step_batches"]["median"] * 1e3,
      d["roofline"]["mean_launch_ms"] * 1e3, d["roofline"]["frac"], d["parity"]["max_abs"]))
PY
