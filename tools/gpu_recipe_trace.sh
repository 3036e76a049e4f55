#!/bin/bash
# Dev: kernel timeline of one forward of a secondary recipe.  usage: bash tools/gpu_recipe_trace.sh <recipe> [<recipe> ...]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for r in "$@"; do
  O=$R/gpurun_out/rt_$r; rm -rf $O; mkdir -p $O
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/recipe_trace.py run $r 2>&1 | grep "host enqueue"
  echo "== $r"; python3 $R/tools/recipe_trace.py report $O
done
