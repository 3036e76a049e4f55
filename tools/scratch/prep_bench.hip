// Dev tool (round 3): role-by-role timing of hicom::query_prep_kernel (cold caches: a 512 MiB memset between launches).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/scratch/prep_bench.hip hicom_amd/csrc/small_ops.hip -o /tmp/prep_bench && /tmp/prep_bench
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../../hicom_amd/csrc/query_prep.hip"

__global__ void null_kernel(int* p) { if (p) *p = 0; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    const int E = 1152, nh = 9, P = 310, hidden = 896;
    auto dalloc = [](size_t n) { void* p; CK(hipMalloc(&p, n)); CK(hipMemset(p, 0x11, n)); return p; };
    PrepParams p{};
    p.g = (const uint16_t*)dalloc(E * 2); p.lq = p.g;
    p.wq = (const uint16_t*)dalloc((size_t)E * E * 2); p.bq = (const uint16_t*)dalloc(E * 2); p.wk = (const uint16_t*)dalloc((size_t)E * E * 2);
    p.kpe = (const float*)dalloc((size_t)E * P * 4);
    p.E = E; p.nh = nh; p.hd = E / nh; p.P = P; p.scale = 0.088f;
    p.qhi = (uint16_t*)dalloc(16 * E * 2); p.qlo = (uint16_t*)dalloc(16 * E * 2); p.pos_a = (float*)dalloc(16 * P * 4); p.pos_stride = P; p.R = nh;
    p.gw0 = (const uint16_t*)dalloc((size_t)hidden * E * 2); p.gb0 = (const uint16_t*)dalloc(hidden * 2); p.bo = (const uint16_t*)dalloc(E * 2);
    p.hidden = hidden; p.r0 = (float*)dalloc(hidden * 4);
    char* state = (char*)dalloc(E * 8 + 64);
    void* trash = dalloc((size_t)512 << 20);
    p.gran = (unsigned long long*)(state + 64); p.state = (unsigned*)state;
    const int NQ = (E + 4 * kQRows - 1) / (4 * kQRows), NR = (hidden + 4 * kRRows - 1) / (4 * kRRows), NF = nh * (E / 128), NP = nh * ((P + 63) / 64);
    struct V { const char* name; int nq, nr, nf, np; };
    V vs[] = {{"full", NQ, NR, NF, NP}, {"q_proj only", NQ, 0, 0, 0}, {"q_proj + r0", NQ, NR, 0, 0}, {"q_proj + fold-w", NQ, 0, NF, 0},
              {"q_proj + fold-w + fold-pos", NQ, 0, NF, NP}, {"r0 only", 0, NR, 0, 0}};
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    {   // event-to-event cost of an empty launch (subtract from the numbers below)
        std::vector<float> ts;
        for (int it = 0; it < 30; ++it) {
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(null_kernel, dim3(190), dim3(256), 0, 0, (int*)nullptr);
            CK(hipEventRecord(b, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("empty kernel, 190 workgroups: median %.2f us  min %.2f\n", ts[ts.size() / 2], ts[0]);
    }
    for (int cold = 0; cold < 2; ++cold)
        for (auto& v : vs) {
            p.nq_wg = v.nq; p.nr_wg = v.nr; p.nf_wg = v.nf; p.np_wg = v.np;
            std::vector<float> ts;
            for (int it = 0; it < 30; ++it) {
                CK(hipMemset(state, 0, E * 8 + 64));
                if (cold) CK(hipMemsetAsync(trash, it, (size_t)512 << 20, 0));
                CK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(query_prep_kernel, dim3(v.nq + v.nr + v.nf + v.np), dim3(256), 0, 0, p);
                CK(hipEventRecord(b, 0));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            unsigned st[3]; CK(hipMemcpy(st, state, 12, hipMemcpyDeviceToHost));
            printf("%-5s %-28s grid %3d: median %6.2f us  min %6.2f  (give-ups %u)\n", cold ? "cold" : "warm", v.name, v.nq + v.nr + v.nf + v.np,
                   ts[ts.size() / 2], ts[0], st[2]);
        }
    return 0;
}
