#!/bin/bash
# Dev: on a box that reproduces the eager-backward forward transient (torch events: the susceptible setting), which switch makes it go away?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-flake_hunt}; mkdir -p $O; cd $R
export HICOM_EVENT_NOFENCE=0
run() { env "$@" timeout 900 python3 tools/flake_loop2.py ${REPS:-8} 2>&1 | grep -E "FAIL|FLAKE_LOOP2" | cut -c1-260; }
REPS=10 run A=1 | tee $O/base.txt
if grep -q "^FAIL" $O/base.txt; then
  REPS=20 run FLAKE_ONE_STREAM=1 | tee $O/one_stream.txt
  REPS=20 run FLAKE_NO_GSTORE=1 | tee $O/no_gstore.txt
  REPS=20 run FLAKE_NO_CTX=1 | tee $O/no_ctx.txt
  REPS=10 run A=1 | tee $O/base2.txt
fi
