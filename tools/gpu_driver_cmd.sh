#!/bin/bash
# Dev: the driver's own command on a fresh lease (full line), then twice more without the extras; then the world-1 distributed branch.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-drv}; rm -rf $O; mkdir -p $O; cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/full.json 2> $O/full.err
python3 - $O/full.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("driver cmd: %.2f us  median %.2f  pipelined %.2f  ring %.2f (%.3f)  whole-step frac %.3f" % (d["ms_per_step"] * 1e3, d["ms_per_step_batches"]["median"] * 1e3,
      d["ms_per_step_pipelined"]["median"] * 1e3, d["roofline"]["mean_launch_ms"] * 1e3, d["roofline"]["frac"], d["whole_step_hbm_frac"]))
print(" pre_warmup", d["pre_warmup"]); print(" host", d["timed_region_host"]); print(" parity", d["parity"]["max_abs"])
print(" cpu", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
print(" secondary", {k: (round(v["ms_per_forward"], 3), round(v.get("train_step_ms", 0), 3)) for k, v in d["secondary"].items()})
n = d["neighbours"]; print(" head", round(n["siglip_head_projection"]["ms"], 3), round(n["siglip_head_projection"]["tflops"]), " train", n["train_step"])
PY
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-extras: %.2f us' % (d['ms_per_step']*1e3))"; done
HICOM_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('world-1 distributed branch: headline (pipelined) %.2f us  joined %.2f  world_size %s' % (d['ms_per_step']*1e3, d['ms_per_step_joined']['median']*1e3, d['world_size']))"
