"""Dev tool: N training steps of one recipe at C2 (run under rocprofv3 by tools/gpu_prof_cmd.sh for the per-kernel table).
usage: train_kernels.py adaptkv|release|off|coarse|fine [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "adaptkv"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
R = {"off": ("local43_global32", None), "coarse": ("local43_global32", "coarse"), "fine": ("local43_global32", "fine"),
     "adaptkv": ("local43_adaptkv_global32", "direct"), "release": ("local43_global32_coarse", "direct")}
ptype, guide = R[name]
gen = torch.Generator(device=dev).manual_seed(3)
ff = torch.randn(64, 27, 27, 1152, device=dev, generator=gen).bfloat16(); fe = torch.randn(64, 27, 27, 1152, device=dev, generator=gen).bfloat16()
g = torch.randn(64, 1152, device=dev).bfloat16() if guide == "fine" else torch.randn(1152, device=dev).bfloat16()
cfg = bench.release_config(896, 64); cfg.mm_projector_type = ptype; cfg.use_guide = guide
m = bench.make_projector(cfg, dev).train()
out = m(ff, fe, g, "video", None)
cot = torch.randn(out.shape, device=dev).to(out.dtype)
def step():
    m.zero_grad(set_to_none=True)
    m(ff, fe, g, "video", None).backward(cot)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"{name}: {steps + 3} steps + 1 forward in the trace; train step {(time.perf_counter() - t0) / steps * 1e3:.2f} ms")
