"""bench.py's own rank launcher (`python bench.py --gpus N` without torchrun): relay of rank 0's JSON line and the failure
paths, driven with a stub child on CPU -- the driver's first real 8-GPU run must not be the first time this code executes."""
import argparse
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / "stub_child.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def _args(n):
    return argparse.Namespace(gpus=n)


def test_relays_rank0_json_line_and_sets_rank_env(tmp_path, capsys):
    cmd = _stub(tmp_path, """
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        print("RCCL banner noise")                       # only rank 0's stdout is a pipe; lines not starting with '{' are dropped
        if r == 0:
            print(json.dumps({"metric": "m", "n_gpus": w, "ranks_seen": w}))
            print("{\\"metric\\": \\"m\\", \\"n_gpus\\": %d, \\"last\\": true}" % w)
    """)
    rc = bench.spawn_ranks(_args(4), child_cmd=cmd, n_devices=4, timeout_s=60)
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1
    assert json.loads(out[0]) == {"metric": "m", "n_gpus": 4, "last": True}      # the LAST JSON line of rank 0


def test_a_failing_rank_fails_the_run(tmp_path, capsys):
    cmd = _stub(tmp_path, """
        import os, sys
        if os.environ["RANK"] == "1":
            sys.exit(7)
        if os.environ["RANK"] == "0":
            print('{"metric": "m"}')
    """)
    rc = bench.spawn_ranks(_args(2), child_cmd=cmd, n_devices=2, timeout_s=60)
    assert rc == 7 and capsys.readouterr().out.strip() == ""


def test_no_result_line_and_too_few_devices(tmp_path, capsys):
    cmd = _stub(tmp_path, "print('no json here')\n")
    assert bench.spawn_ranks(_args(2), child_cmd=cmd, n_devices=2, timeout_s=60) == 5
    assert bench.spawn_ranks(_args(8), child_cmd=cmd, n_devices=1, timeout_s=60) == 3
    assert capsys.readouterr().out.strip() == ""


def test_a_hung_rank_is_killed_and_reported(tmp_path, capsys):
    cmd = _stub(tmp_path, """
        import os, time
        if os.environ["RANK"] == "0":
            print('{"metric": "m"}', flush=True)
        time.sleep(60 if os.environ["RANK"] == "1" else 0)
    """)
    rc = bench.spawn_ranks(_args(2), child_cmd=cmd, n_devices=2, timeout_s=2)
    assert rc == 4 and capsys.readouterr().out.strip() == ""
