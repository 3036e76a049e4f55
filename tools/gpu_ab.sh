#!/bin/bash
# Dev: A/B of environment switches on the same box.  usage: bash tools/gpu_ab.sh <tag> "<pytest -k expr or empty>" "ENV1=.. ENV2=.." "ENVX=.." ...
TAG=$1; KEXPR=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
if [ -n "$KEXPR" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -q -x -k "$KEXPR" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  tail -4 $O/pytest.log | cut -c1-200
fi
for rep in 1 2; do
  i=0
  for envs in "$@"; do
    i=$((i+1))
    env $envs timeout 300 python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-secondary 2> $O/bench_${i}_$rep.err | grep '^{' > $O/bench_${i}_$rep.json
    python3 - "$envs" $O/bench_${i}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-44s step %.2f us  median %.2f  same-buf %.2f  ring %.2f us (%.3f)" % (sys.argv[1], d["ms_per_step"] * 1e3, d["ms_per_step_batches"]["median"] * 1e3,
      d["ms_per_step_same_buffers"]["median"] * 1e3, d["roofline"]["mean_launch_ms"] * 1e3, d["roofline"]["frac"]))
PY
  done
done
