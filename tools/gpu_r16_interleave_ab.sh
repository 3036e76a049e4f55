#!/bin/bash
# Dev A/B: readout tiles with the next stage's DMA pieces issued BETWEEN the MFMAs (product) against all of them right behind the barrier
# (-DHICOM_R16_SERIAL_ISSUE build: rounds 2-6), same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r16il}; mkdir -p $O; cd $R
python3 -c "
from hicom_amd import build_native as bn
bn.build(lib_path='/tmp/libhicom_serial.so', extra_flags=('-DHICOM_R16_SERIAL_ISSUE',), verbose=False)"
timeout 900 python3 -m pytest tests -m gpu -q -x -k "3584 or wide or readout16 or width or release or golden or role or chain" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
timeout 300 python3 tools/r16_hot_trace.py 2>&1 | grep -E "hot|cold" | tee $O/hot_trace.txt
one() { python3 bench.py --gpus 1 --steps 1000 --warmup 200 --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: step %.2f us  ring %.2f (%.3f)' % (d['ms_per_step']*1e3, d['roofline']['mean_launch_ms']*1e3, d['roofline']['frac']))"; }
for i in 1 2 3; do
  HICOM_NATIVE_LIB=/tmp/libhicom_serial.so one "serial issue"
  one "interleaved "
  HICOM_NATIVE_LIB=/tmp/libhicom_serial.so python3 tools/c4_step.py 2000 32 2>/dev/null | tail -1 | cut -c1-60 | sed "s/^/serial issue  /"
  python3 tools/c4_step.py 2000 32 2>/dev/null | tail -1 | cut -c1-60 | sed "s/^/interleaved   /"
done | tee $O/ab.txt
