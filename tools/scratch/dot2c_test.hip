// Dev tool (round 3): semantics of v_dot2c_f32_bf16 against a plain fp32 evaluation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
__global__ void k(const uint32_t* a, const uint32_t* b, float* o) {
    const int i = threadIdx.x;
    float acc = 0.f;
    const unsigned ai = a[i], bi = b[i];
    asm("v_dot2c_f32_bf16_e32 %0, %1, %2" : "+v"(acc) : "v"(ai), "v"(bi));
    o[i] = acc;
}
static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    uint32_t ha[64], hb[64]; float ho[64];
    for (int i = 0; i < 64; ++i) { ha[i] = 0x3f800000u + (i << 7) + (((0x4000 + 37 * i) & 0xffff)); hb[i] = (0xbf00u + i) << 16 | (0x3e80u + 3 * i); }
    uint32_t *a, *b; float* o;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 256);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o);
    hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 64; ++i) {
        const double want = (double)bf(ha[i] & 0xffff) * bf(hb[i] & 0xffff) + (double)bf(ha[i] >> 16) * bf(hb[i] >> 16);
        worst = fmax(worst, fabs(want - ho[i]) / fmax(1e-30, fabs(want)));
        if (i < 4) printf("lane %d: got %.8g want %.8g\n", i, ho[i], want);
    }
    printf("worst relative error %.3g\n", worst);
    return 0;
}
