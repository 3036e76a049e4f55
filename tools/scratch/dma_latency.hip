// Dev probe (round 6, verdict r5 #7: "where the 3 us go" of DESIGN.md §3.4): request -> landing time of ONE 1-KiB LDS-DMA piece
// (global_load_lds_dwordx4, 64 lanes x 16 B) as a wave of a GEMM staging loop sees it, by SOURCE of the bytes and by LOAD:
//   source  l2    every wave walks the same 2-MiB region (resident in every XCD's 4-MiB L2 after the warm-up pass)
//           mall  pseudo-random 1-KiB pieces of a 128-MiB region (beyond the L2s, inside the 256-MiB Infinity Cache after the warm-up)
//           hbm   every wave streams its own contiguous range of a 4-GiB buffer, each byte once per launch
//   load    D pieces in flight per wave x 4 waves x WG workgroups per CU (dense16's staging keeps 4 WG/CU x 32 KiB = 128 KiB in flight per CU)
// A wave keeps D pieces in flight: it waits for its OLDEST piece (counted vmcnt), stamps s_memrealtime (100 MHz), issues the next piece
// into the freed slot.  Sample = landing stamp - issue stamp of that piece.  Samples go to LDS (a global store would count in vmcnt).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/scratch/dma_latency.hip -o /tmp/dma_latency && /tmp/dma_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

constexpr int kSamples = 256;                    // per wave

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int D>
__global__ __launch_bounds__(256) void probe(const char* base, unsigned long long region, int mode, unsigned* lat, unsigned* xcc_of) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* slot0 = lds + wave * D * 1024;
    unsigned* smp = reinterpret_cast<unsigned*>(lds + 4 * D * 1024) + wave * kSamples;
    const unsigned long long gw = (unsigned long long)blockIdx.x * 4 + wave, nwaves = (unsigned long long)gridDim.x * 4;
    // (regions and per-wave ranges are powers of two: masks, no division -- the address of the NEXT piece is computed before the stamp, so a
    // sample holds the memory system's time, not the probe's address arithmetic)
    const unsigned long long pieces = region >> 10, pmask = pieces - 1, per_wave = pieces / nwaves, wmask = per_wave - 1;
    auto src = [&](unsigned long long k) -> const char* {
        unsigned long long piece;
        if (mode == 0) piece = (gw * 977ULL + k) & pmask;                       // a small region every wave walks
        else if (mode == 1) piece = mix(gw * 0x9E3779B97F4A7C15ULL + k) & pmask;    // random pieces of a cache-sized region
        else piece = gw * per_wave + (k & wmask);                               // this wave's own contiguous range
        return base + (piece << 10) + lane * 16;
    };
    auto issue = [&](const char* a, int d) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                         (__attribute__((address_space(3))) void*)(slot0 + d * 1024), 16, 0, 0);
    };
    unsigned long long t_issue[D];
    unsigned long long k = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const char* a = src(k++);
        asm volatile("" : "+v"(a));
        t_issue[d] = __builtin_amdgcn_s_memrealtime();
        issue(a, d);
    }
    const char* nxt = src(k++);
    int n = 0;
    for (int it = 0; it < kSamples / D + 4; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            wait_vm<D - 1>();                                                   // the oldest piece (slot d) has landed
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if (it >= 4 && n < kSamples) {                                      // (the first rounds: ramp)
                if (lane == 0) smp[n] = (unsigned)(t - t_issue[d]);
                ++n;
            }
            asm volatile("" : "+v"(nxt));
            t_issue[d] = __builtin_amdgcn_s_memrealtime();
            issue(nxt, d);
            nxt = src(k++);
        }
    }
    wait_vm<0>();
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * kSamples; i += 256) lat[(unsigned long long)blockIdx.x * 4 * kSamples + i] = reinterpret_cast<unsigned*>(lds + 4 * D * 1024)[i];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_of[blockIdx.x] = x & 15;
    }
}

template <int D>
static void run(const char* name, const char* buf, unsigned long long region, int mode, int grid, unsigned* dlat, unsigned* dx) {
    const size_t lds = 4 * D * 1024 + 4 * kSamples * 4;
    hipFuncSetAttribute((const void*)probe<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {                                          // (the first launches warm the caches of the l2 / mall modes)
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<D>, dim3(grid), dim3(256), lds, 0, buf, region, mode, dlat, dx);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned> h((size_t)grid * 4 * kSamples), hx(grid);
    hipMemcpy(h.data(), dlat, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hx.data(), dx, hx.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned> all(h);
    std::sort(all.begin(), all.end());
    auto q = [&](const std::vector<unsigned>& v, double f) { return v.empty() ? 0.0 : v[(size_t)(f * (v.size() - 1))] / 100.0; };
    const double pieces = (double)grid * 4 * (kSamples / D + 4) * D + (double)grid * 4 * D;
    printf("%-5s D=%d  %4d WGs (%d/CU)  in flight/CU %4d KiB | request->landing us: p10 %.2f  median %.2f  p90 %.2f  p99 %.2f | launch %.1f us, %.0f GB/s per CU | per-XCC median:",
           name, D, grid, grid >= 256 ? grid / 256 : 0, (grid >= 256 ? grid / 256 : 1) * 4 * D, q(all, 0.10), q(all, 0.50), q(all, 0.90), q(all, 0.99),
           ms * 1e3, pieces * 1024 / (ms * 1e-3) / 1e9 / (grid >= 256 ? 256 : 1));
    for (int x = 0; x < 8; ++x) {
        std::vector<unsigned> v;
        for (int b = 0; b < grid; ++b)
            if ((int)hx[b] == x) v.insert(v.end(), h.begin() + (size_t)b * 4 * kSamples, h.begin() + (size_t)(b + 1) * 4 * kSamples);
        std::sort(v.begin(), v.end());
        printf(" %.2f", q(v, 0.5));
    }
    printf("\n");
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const unsigned long long big = 4ULL << 30;
    char* buf = nullptr;
    if (hipMalloc(&buf, big) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, big);
    unsigned *dlat, *dx;
    hipMalloc(&dlat, (size_t)1024 * 4 * kSamples * 4);
    hipMalloc(&dx, 1024 * 4);
    struct { const char* name; unsigned long long region; int mode; } srcs[] = {{"l2", 2ULL << 20, 0}, {"mall", 128ULL << 20, 1}, {"hbm", big, 2}};
    printf("LDS-DMA piece (1 KiB = one wave instruction) request -> landing, s_memrealtime stamps (10-ns ticks), 256 samples per wave\n");
    for (auto& s : srcs) {
        run<1>(s.name, buf, s.region, s.mode, 1, dlat, dx);            // one workgroup on an idle chip, one piece in flight per wave: the bare latency
        run<1>(s.name, buf, s.region, s.mode, 256, dlat, dx);          // one workgroup per CU, 4 KiB in flight per CU
        run<8>(s.name, buf, s.region, s.mode, 256, dlat, dx);          // 32 KiB in flight per CU
        run<2>(s.name, buf, s.region, s.mode, 1024, dlat, dx);         // 4 WG/CU, 32 KiB in flight per CU
        run<8>(s.name, buf, s.region, s.mode, 1024, dlat, dx);         // 4 WG/CU, 128 KiB in flight per CU: dense16's staging load
    }
    return 0;
}
