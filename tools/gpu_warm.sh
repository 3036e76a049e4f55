#!/bin/bash
# Dev (round 5): the driver's own bench command on a fresh lease, three fresh processes, then the warm-up probe, then the driver's
# command under the kernel tracer (per-kernel start / end stamps of the timed region).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-warm}; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $O/drv_$i.json 2> $O/drv_$i.err
  python3 -c "
import json; d=json.loads([l for l in open('$O/drv_$i.json') if l.startswith('{')][-1]); print('driver cmd (no extras) run $i: %.2f us' % (d['ms_per_step']*1e3), d.get('pre_warmup'))"
done
python3 tools/warm_probe.py > $O/probe.json 2> $O/probe.err
python3 - <<PY
import json
d = json.load(open("$O/probe.json"))
s = d.get("stamps_first", [])
print("stamps first 400 (us), every 20th mean:", [round(sum(s[i:i+20])/20, 1) for i in range(0, len(s), 20)])
w = d["windows"]
print("windows (t_ms, us/step):", [w[i] for i in (0,1,2,3,4,5,7,10,15,20,30,50,75,100,149)])
print("after_idle", d["after_idle"])
print("after_queue", d["after_queue"])
print("k_sweep", d["k_sweep"])
sl = d["stamps_late"]; print("stamps late mean", sum(sl)/len(sl), "final", d["final_windows"])
PY
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $O/trace.log 2>&1
python3 tools/region_trace.py $O/trace 20 | tee $O/region_trace.txt
