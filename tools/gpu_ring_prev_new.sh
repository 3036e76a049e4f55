#!/bin/bash
# Dev: ring kernel A/B on ONE box -- the tree's fused_ring.hip against tools/scratch/fused_ring_prev.hip.txt (built into /tmp/prev/libprev.so).
R=$GRAFT_REPO_ROOT; cd $R
rm -rf /tmp/prev; mkdir -p /tmp/prev/hicom_amd; cp -r hicom_amd/csrc /tmp/prev/hicom_amd/csrc; cp -r include /tmp/prev/include
cp tools/scratch/fused_ring_prev.hip.txt /tmp/prev/hicom_amd/csrc/fused_ring.hip
( cd /tmp/prev/hicom_amd/csrc && for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c $f -o ${f%.hip}.o & done; wait; hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/prev/libprev.so *.o ) > /tmp/prev/build.log 2>&1 || tail -5 /tmp/prev/build.log
for rep in 1 2; do
  python3 tools/ring_bench.py
  HICOM_NATIVE_LIB=/tmp/prev/libprev.so python3 tools/ring_bench.py
done
