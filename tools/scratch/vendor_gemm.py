"""Dev probe: what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on the head-projection shapes (NT form, plain epilogue):
the known-good reference on this hardware for hicom_dense16_gemm_fwd's numbers (cdna_hip_programming.md §5.4 rule 10)."""
import time, torch
def run(M, N, K, dt):
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * 0.03).to(dt)
    f = lambda: torch.mm(a, w.t())
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"{str(dt):16s} M={M} N={N} K={K}: {t*1e3:.3f} ms {2.0*M*N*K/t/1e12:.0f} TFLOP/s")
for dt in (torch.float16, torch.bfloat16):
    run(46656, 4352, 1152, dt); run(46656, 1152, 4352, dt); run(46656, 1152, 1152, dt); run(8192, 8192, 8192, dt)
