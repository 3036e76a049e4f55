// Dev: how many 256-thread workgroups with 160 KB of dynamic LDS does the chip run AT ONCE, and where?  Each workgroup stamps
// s_memrealtime at entry and exit and its XCC / SE / CU ids, then spins ~20 us.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
struct Rec { unsigned long long t0, t1; unsigned xcc, hwid; };
__global__ __launch_bounds__(256) void census(Rec* r, int spin_ticks) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        lds[0] = 1;
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(4);
        r[blockIdx.x] = Rec{t0, __builtin_amdgcn_s_memrealtime(), xcc & 15, hw};
    }
}
int main(int argc, char** argv) {
    int lds = argc > 1 ? atoi(argv[1]) : 163840;
    hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    Rec* d; hipMalloc(&d, 4096 * sizeof(Rec));
    for (int n : {240, 248, 256, 264, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(census, dim3(n), dim3(256), lds, 0, d, 2000);   // 2000 ticks of 10 ns = 20 us
            hipDeviceSynchronize();
        }
        std::vector<Rec> h(n); hipMemcpy(h.data(), d, n * sizeof(Rec), hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull; for (auto& x : h) tmin = std::min(tmin, x.t0);
        int early = 0, perx[16] = {0}; std::vector<int> late;
        for (int i = 0; i < n; ++i) { if (h[i].t0 - tmin < 500) { ++early; perx[h[i].xcc]++; } else late.push_back(i); }
        printf("grid %d, lds %d: %d workgroups started within 5 us; per XCC:", n, lds, early);
        for (int x = 0; x < 8; ++x) printf(" %d", perx[x]);
        printf("; late blocks:");
        for (size_t i = 0; i < late.size() && i < 12; ++i) printf(" %d(+%.1fus,xcc%u)", late[i], (h[late[i]].t0 - tmin) / 100.0, h[late[i]].xcc);
        printf("\n");
    }
    return 0;
}
