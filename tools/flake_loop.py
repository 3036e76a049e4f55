"""Dev: repeats tests/test_gpu_backward.py::test_graph_backward_follows_weight_updates_between_steps IN ONE PROCESS (the once-seen mismatch of
round 5 happened inside a full pytest run, i.e. with plan caches, stores and the allocator already warm from other cases), interleaving the
recipes so that every case runs behind every other.  usage: python3 tools/flake_loop.py <reps> [case,case,...]
Prints one line per failure (the test's own diagnostics) and a summary line `FLAKE_LOOP ... failures=N of M`."""
import os, sys, time, traceback
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import test_gpu_backward as tb

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["G7_fine", "G1_direct_T8", "G6_coarse", "G5_adaptkv"]
weights = {"G7_fine": 4}                       # the case that failed once runs more often
runs = fails = 0
t0 = time.time()
for rep in range(reps):
    for name in names:
        for how in ("add_", "data_copy_"):
            for _ in range(weights.get(name, 1)):
                runs += 1
                try:
                    tb.test_graph_backward_follows_weight_updates_between_steps(name, how)
                except AssertionError as e:
                    fails += 1
                    print(f"FAIL rep {rep} {name} {how}: {e}", flush=True)
                except Exception:
                    fails += 1
                    print(f"ERROR rep {rep} {name} {how}:\n{traceback.format_exc()}", flush=True)
torch.cuda.synchronize()
print(f"FLAKE_LOOP device {torch.cuda.get_device_name(0)} reps {reps} cases {names}: failures={fails} of {runs} runs in {time.time() - t0:.0f} s", flush=True)
