#!/bin/bash
# first GPU pass of round 2: parity suite, honest bench line, joined-forward timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ra; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_grad_mode_builds_a_graph_or_refuses > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py --no-secondary > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-1500 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 200 --warmup 20 --no-extras > $O/trace.log 2>&1
cd $R
python3 tools/timeline_all.py $(ls $O/trace/*/*kernel_trace.csv | head -1) 24
