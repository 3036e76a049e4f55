// Dev probe (round 6): per-CU LDS-DMA fill rate from the XCD's L2 by ACCESS PATTERN.  A readout tile stages, per 64-deep K step, 160 operand rows
// x 128 B at a row stride of 2 K bytes (2304 B at K = 1152, 7168 B at K = 3584): 8 rows per wave instruction.  fill_forms.hip's 120-127 GB/s per CU
// are 1-KiB CONTIGUOUS pieces.  Same bytes, same L2-resident panel, two layouts:
//   rows    [R][K] row-major: piece = 8 rows x 128 B, the next K step 128 B further along the same rows            (what the GEMM reads today)
//   tiled   [K / 64][R][64]: a K step's rows are contiguous (R x 128 B): piece = 1 KiB contiguous                   (a pre-tiled operand)
// Each workgroup (256 threads, 1 per CU unless noted) walks a 160-row panel over all K steps, D = 7 stages (of 20 pieces) in flight, repeatedly.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/scratch/fill_strided.hip -o /tmp/fill_strided && /tmp/fill_strided
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int kRows = 160, kPW = 5, kDepth = 7;      // rows of a panel, pieces per wave and stage (20 / 4), stages in flight

template <bool TILED>
__global__ __launch_bounds__(256) void fill(const char* base, int R, int K, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ns = K / 64;
    const int r0 = (int)((blockIdx.x * 37u) % (unsigned)(R - kRows));
    // piece pi (0..19) of a stage = rows [8 pi, 8 pi + 8) x 128 B; wave w takes pieces w, w + 4, ...
    const char* src[kPW];
#pragma unroll
    for (int i = 0; i < kPW; ++i) {
        const int pi = wave + 4 * i, row = r0 + 8 * pi + (lane >> 3), ch = lane & 7;
        src[i] = TILED ? base + ((long)row * 64 + ch * 8) * 2 : base + ((long)row * K + ch * 8) * 2;
    }
    const long step = TILED ? (long)R * 128 : 128;     // bytes from one K step to the next
    auto issue = [&](int s, int slot) {
#pragma unroll
        for (int i = 0; i < kPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + s * step),
                                             (__attribute__((address_space(3))) void*)(lds + slot * 20480 + (wave + 4 * i) * 1024), 16, 0, 0);
    };
    const int total = ns * passes;
    for (int s = 0; s < kDepth; ++s) issue(s % ns, s);
    int slot = 0;
    for (int s = 0; s + kDepth < total; ++s) {
        wait_vm<(kDepth - 1) * kPW>();
        issue((s + kDepth) % ns, slot);
        slot = slot + 1 == kDepth ? 0 : slot + 1;
    }
    wait_vm<0>();
    __syncthreads();
    if (reinterpret_cast<unsigned*>(lds)[threadIdx.x] == 0x12345678u) sink[0] = 1;
}

template <bool TILED>
static void run(const char* name, const char* buf, int R, int K, int grid, unsigned* sink) {
    const int passes = 64;
    hipFuncSetAttribute((const void*)fill<TILED>, hipFuncAttributeMaxDynamicSharedMemorySize, kDepth * 20480);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fill<TILED>), dim3(grid), dim3(256), kDepth * 20480, 0, buf, R, K, passes, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)grid * (K / 64) * passes * 20480.0;
    printf("%-6s R=%4d K=%4d (row stride %5d B, panel %5.2f MB)  %3d WGs  launch %7.1f us  %6.1f GB/s per CU\n", name, R, K, 2 * K, R * (double)K * 2 / 1e6, grid, best * 1e3,
           bytes / (best * 1e-3) / 1e9 / grid);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    char* buf = nullptr;
    hipMalloc(&buf, 64 << 20);
    hipMemset(buf, 1, 64 << 20);
    unsigned* sink;
    hipMalloc(&sink, 64);
    printf("per-CU LDS-DMA fill rate of a 160-row operand panel walked along K (7 stages of 20 KiB in flight), by layout\n");
    for (int grid : {196, 256}) {
        run<false>("rows", buf, 896, 1152, grid, sink);
        run<true>("tiled", buf, 896, 1152, grid, sink);
        run<false>("rows", buf, 448, 3584, grid, sink);
        run<true>("tiled", buf, 448, 3584, grid, sink);
        run<false>("rows", buf, 1296, 896, grid, sink);
        run<true>("tiled", buf, 1296, 896, grid, sink);
    }
    return 0;
}
