#!/bin/bash
# Dev: A/B of environment switches on one box for the non-release recipes.  usage: bash tools/gpu_modes_ab.sh "off coarse" "ENVA=.." "ENVB=.." ...
MODES=$1; shift
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for envs in "$@"; do
    echo "== $envs"; env $envs timeout 300 python3 tools/mode_time.py $MODES 2>/dev/null
  done
done
