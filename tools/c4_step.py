"""Dev tool: the compressor at BASELINE configs[3]'s shape (32 frames, hidden 3584 = Qwen2.5-7B width) as a loop for rocprofv3 --kernel-trace --stats;
prints the un-profiled step and its fraction of the whole-step HBM roofline."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda", 0)
T, HID = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 3584      # argv[2]: frames (default: configs[3]'s 32)
m = bench.make_projector(bench.release_config(HID, T), dev)
sets = [(torch.randn(T, 27, 27, 1152, device=dev).bfloat16(), torch.randn(T, 27, 27, 1152, device=dev).bfloat16(), torch.randn(1152, device=dev).bfloat16()) for _ in range(4)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
with torch.no_grad():
    for i in range(300):
        a, b, g = sets[i % 4]; m(a, b, g, "video", None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        a, b, g = sets[i % 4]; out = m(a, b, g, "video", None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
alg = 3359232 * T + 78623744 + out.shape[0] * HID * 2 + 2304      # frames + weights (H = 3584) + output + guide
print("C4 compressor step (%d frames) %.2f us, %d tokens; algorithmic bytes %.1f MB -> %.3f of the 8 TB/s whole-step roofline" % (T, dt * 1e6, out.shape[0], alg / 1e6, alg / dt / 8e12))
