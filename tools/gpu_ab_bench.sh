#!/bin/bash
# Dev (round 5): the driver's command on a fresh lease, this round's bench.py against round 4's (_bench_r04.py), alternating.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-ab}; rm -rf $O; mkdir -p $O
cd $R
i=0
for b in bench.py _bench_r04.py bench.py _bench_r04.py bench.py _bench_r04.py; do
  i=$((i+1))
  extra="--no-extras"; [ $i -le 2 ] && extra=""
  [ -f $b ] || continue
  python3 $b --gpus 1 --steps 20 --warmup 5 $extra > $O/run_$i.json 2> $O/run_$i.err
  python3 -c "
import json; d=json.loads([l for l in open('$O/run_$i.json') if l.startswith('{')][-1])
print('$b $extra: %.2f us' % (d['ms_per_step']*1e3), 'median', (d.get('ms_per_step_batches') or {}).get('median'), d.get('pre_warmup'), d.get('timed_region_host'))"
done
