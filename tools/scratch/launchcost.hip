// Dev micro-benchmark: host cost of one kernel launch through the three HIP entry points.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
struct P { float* x; int n; long pad[12]; };
__global__ void k(P p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p.n < 0) p.x[0] = 1.f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* x; hipMalloc(&x, 4); P p{x, 1, {}};
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t ev; hipEventCreate(&ev);
    hipFunction_t f; hipGetFuncBySymbol(&f, (const void*)k);
    void* args[] = {&p};
    const int N = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, p);
        hipStreamSynchronize(s);
        double t0 = now(); for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, p); if (i % 16 == 15) hipStreamSynchronize(s); } double t1 = now();
        hipStreamSynchronize(s);
        double t2 = now(); for (int i = 0; i < N; ++i) { hipModuleLaunchKernel(f, 64, 1, 1, 256, 1, 1, 0, s, args, nullptr); if (i % 16 == 15) hipStreamSynchronize(s); } double t3 = now();
        hipStreamSynchronize(s);
        double t4 = now(); for (int i = 0; i < N; ++i) { hipExtLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, nullptr, ev, 0, p); if (i % 16 == 15) hipStreamSynchronize(s); } double t5 = now();
        hipStreamSynchronize(s);
        double t6 = now(); for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, p); hipEventRecord(ev, s); if (i % 16 == 15) hipStreamSynchronize(s); } double t7 = now();
        hipStreamSynchronize(s);
        double t8 = now(); for (int i = 0; i < N; ++i) { hipStreamSynchronize(s); } double t9 = now();
        printf("per launch (incl. 1/16 sync): GGL %.2f us   module %.2f us   ext+event %.2f us   GGL+record %.2f us   (idle sync %.2f us)\n",
               (t1 - t0) / N, (t3 - t2) / N, (t5 - t4) / N, (t7 - t6) / N, (t9 - t8) / N);
    }
    return 0;
}
