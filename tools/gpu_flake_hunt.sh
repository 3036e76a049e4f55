#!/bin/bash
# Dev: on a box that reproduces the eager-backward forward transient (torch events: the susceptible setting), which switch makes it go away?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-flake_hunt}; mkdir -p $O; cd $R
export HICOM_EVENT_NOFENCE=0
run() { env "$@" timeout 900 python3 tools/flake_loop2.py ${REPS:-8} 2>&1 | grep -E "FAIL|FLAKE_LOOP2" | cut -c1-260; }
REPS=10 run A=1 | tee $O/base.txt
if grep -q "^FAIL" $O/base.txt; then
  # (second hunt: does the victim follow the ORDER of the two modules?  ab: the eager module's forward runs right behind the graph module's replay)
  REPS=24 run FLAKE_ORDER=ba | tee $O/order_ba.txt
  REPS=12 run FLAKE_ORDER=ab | tee $O/order_ab.txt
fi
