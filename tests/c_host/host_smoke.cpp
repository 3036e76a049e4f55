// A torch-free host of the C ABI (include/hicom_hip.h): hipMalloc'd buffers, the window-attention operator of the local compressor
// (reference projector.py:544-553, direct guide) called straight from C++, checked against a double-precision loop in this file.
// Built and run by tests/test_gpu_c_host.py:  hipcc -I include host_smoke.cpp -L hicom_amd -lhicom_hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hicom_hip.h"

static uint16_t f2bf(float f) {                      // round-to-nearest-even bf16
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main() {
    if (hicom_abi_version() != HICOM_ABI_VERSION) { printf("ABI mismatch\n"); return 3; }
    const int T = 4, H = 6, W = 6, D = 1152, kt = 4, ks = 3;
    const int N = T * H * W, nw = (T / kt) * (H / ks) * (W / ks), win = kt * ks * ks;
    std::vector<uint16_t> key((size_t)N * D), val((size_t)N * D), q(D);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : key) v = f2bf(rnd());
    for (auto& v : val) v = f2bf(rnd());
    for (auto& v : q) v = f2bf(rnd());
    void *dk, *dv, *dq;
    float* dctx;
    CK(hipMalloc(&dk, key.size() * 2)); CK(hipMalloc(&dv, val.size() * 2)); CK(hipMalloc(&dq, q.size() * 2));
    CK(hipMalloc((void**)&dctx, (size_t)nw * D * 4));
    CK(hipMemcpy(dk, key.data(), key.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, val.data(), val.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice));
    hicom_axis at{T, kt, T / kt, T / kt}, ay{H, ks, H / ks, H / ks}, ax{W, ks, W / ks, W / ks};
    const float scale = 1.0f / sqrtf((float)D);
    int rc = hicom_local_attn_fwd(dk, HICOM_DT_BF16, dv, HICOM_DT_BF16, D, at, ay, ax, dq, HICOM_DT_BF16, 0, scale, 0.0f, 0, dctx, nullptr, nullptr);
    if (rc != HICOM_OK) { printf("hicom_local_attn_fwd failed: %d %s\n", rc, hicom_last_error()); return 4; }
    CK(hipDeviceSynchronize());
    std::vector<float> ctx((size_t)nw * D);
    CK(hipMemcpy(ctx.data(), dctx, ctx.size() * 4, hipMemcpyDeviceToHost));
    // an argument error comes back as a status + message, never as an exception or a crash
    rc = hicom_local_attn_fwd(dk, HICOM_DT_BF16, dv, HICOM_DT_BF16, 1000, at, ay, ax, dq, HICOM_DT_BF16, 0, scale, 0.0f, 0, dctx, nullptr, nullptr);
    if (rc == HICOM_OK || !hicom_last_error()[0]) { printf("bad D was accepted\n"); return 5; }
    double worst = 0;
    for (int w = 0; w < nw; ++w) {
        const int w1 = w % (W / ks), h1 = (w / (W / ks)) % (H / ks), t1 = w / ((W / ks) * (H / ks));
        std::vector<double> sc(win);
        std::vector<long> tok(win);
        double mx = -1e300;
        for (int i = 0; i < win; ++i) {
            const int t2 = i / (ks * ks), r = i % (ks * ks), h2 = r / ks, w2 = r % ks;
            tok[i] = ((long)(t1 * kt + t2) * H + (h1 * ks + h2)) * W + (w1 * ks + w2);
            double d = 0;
            for (int c = 0; c < D; ++c) d += (double)bf2f(q[c]) * bf2f(key[tok[i] * D + c]);
            sc[i] = d * scale;
            mx = sc[i] > mx ? sc[i] : mx;
        }
        double sum = 0;
        for (int i = 0; i < win; ++i) { sc[i] = exp(sc[i] - mx); sum += sc[i]; }
        for (int c = 0; c < D; ++c) {
            double a = 0;
            for (int i = 0; i < win; ++i) a += sc[i] / sum * bf2f(val[tok[i] * D + c]);
            const double e = fabs(a - ctx[(size_t)w * D + c]);
            worst = e > worst ? e : worst;
        }
    }
    printf("windows %d, max-abs vs double loop %.3e\n", nw, worst);
    if (!(worst <= 1e-4)) return 6;
    printf("C-HOST OK\n");
    return 0;
}
