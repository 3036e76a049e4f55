// Dev probe (round 6): per-CU fill rate from the XCD's L2 by STAGING FORM -- is the 70 GB/s per CU that bounds a readout tile's K loop
//  a property of the LDS-DMA path (global_load_lds_dwordx4) or of the CU's vector-memory path as such?
//   dma      global_load_lds_dwordx4 into an LDS ring (what readout16 / dense16 do)
//   reg      global_load_dwordx4 into registers, consumed by an XOR (no LDS)
//   reg+lds  global_load_dwordx4 into registers, then ds_write_b128 into the ring (register staging)
// Every wave walks 1-KiB pieces of a 2-MiB region (L2-resident after the first pass) with D pieces in flight; WG = 256 threads.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/scratch/fill_forms.hip -o /tmp/fill_forms && /tmp/fill_forms
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D, int FORM>
__global__ __launch_bounds__(256) void fill(const char* base, unsigned long long region, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* slot0 = lds + wave * D * 1024;
    const unsigned long long gw = (unsigned long long)blockIdx.x * 4 + wave;
    const unsigned long long pmask = (region >> 10) - 1;
    unsigned long long k = 0;
    auto src = [&](unsigned long long kk) -> const char* { return base + (((gw * 977ULL + kk) & pmask) << 10) + lane * 16; };
    uint4 acc = {0, 0, 0, 0};
    if (FORM == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src(k++),
                                             (__attribute__((address_space(3))) void*)(slot0 + d * 1024), 16, 0, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const char* a = src(k++);
                wait_vm<D - 1>();
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                                 (__attribute__((address_space(3))) void*)(slot0 + d * 1024), 16, 0, 0);
            }
        }
        wait_vm<0>();
    } else {
        // (inline asm: the compiler's own schedule of D plain loads collapses to one or two in flight; the wait names the register it frees so
        // that the consumer cannot move in front of it)
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u r[D];
        auto gld = [&](v4u& dst, const char* a) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(a) : "memory"); };
#pragma unroll
        for (int d = 0; d < D; ++d) gld(r[d], src(k++));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const char* a = src(k++);
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[d]) : "n"(D - 1) : "memory");
                if (FORM == 1) {
                    acc.x ^= r[d].x; acc.y ^= r[d].y; acc.z ^= r[d].z; acc.w ^= r[d].w;
                } else {
                    *reinterpret_cast<v4u*>(slot0 + d * 1024 + lane * 16) = r[d];
                }
                gld(r[d], a);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < D; ++d) { asm volatile("" : "+v"(r[d])); acc.x ^= r[d].x; acc.y ^= r[d].y; acc.z ^= r[d].z; acc.w ^= r[d].w; }
    }
    __syncthreads();
    if (FORM != 1) acc.x ^= reinterpret_cast<unsigned*>(lds)[threadIdx.x];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;         // (keeps the loads alive)
}

template <int D, int FORM>
static void run(const char* name, const char* buf, int grid, unsigned* sink) {
    const size_t lds = 4 * D * 1024;
    const int iters = 2048 / D;                                                // 2 MiB per wave
    hipFuncSetAttribute((const void*)fill<D, FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fill<D, FORM>), dim3(grid), dim3(256), lds, 0, buf, 2ULL << 20, iters, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)grid * 4 * (iters + 1) * D * 1024;
    printf("%-8s D=%d  %4d WGs (%d/CU)  %5d KiB in flight/CU  launch %7.1f us  %6.1f GB/s per CU  (%.1f TB/s chip)\n", name, D, grid, grid / 256,
           grid / 256 * 4 * D, best * 1e3, bytes / (best * 1e-3) / 1e9 / 256, bytes / (best * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    char* buf = nullptr;
    hipMalloc(&buf, 4 << 20);
    hipMemset(buf, 1, 4 << 20);
    unsigned* sink;
    hipMalloc(&sink, 64);
    printf("per-CU fill rate from the XCD's L2 (2-MiB region every wave walks) by staging form, 256-thread workgroups\n");
    for (int grid : {256, 512, 1024}) {
        if (grid == 256) { run<8, 0>("dma", buf, 256, sink); run<8, 1>("reg", buf, 256, sink); run<8, 2>("reg+lds", buf, 256, sink); }
        if (grid == 512) { run<8, 0>("dma", buf, 512, sink); run<8, 1>("reg", buf, 512, sink); run<8, 2>("reg+lds", buf, 512, sink); }
        if (grid == 1024) { run<8, 0>("dma", buf, 1024, sink); run<8, 1>("reg", buf, 1024, sink); run<8, 2>("reg+lds", buf, 1024, sink); }
    }
    run<4, 1>("reg", buf, 1024, sink);
    run<16, 1>("reg", buf, 512, sink);
    run<16, 0>("dma", buf, 512, sink);
    return 0;
}
