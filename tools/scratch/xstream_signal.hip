// Dev probe (round 6): what does it cost the PRODUCER stream to tell another stream "kernel i is done", per iteration, by mechanism?
// main stream: N iterations of [work kernel (~10 us, 256 workgroups) | signal]; comm stream: N x [wait | tiny kernel].  Reported: main-stream time
// per iteration (hipEvent pair around the whole loop on the main stream) minus the no-signal loop's.
//   event      hipEventRecord(ev[i], main) + hipStreamWaitEvent(comm, ev[i])
//   stop-event hipExtLaunchKernelGGL(..., stopEvent = ev[i]) + hipStreamWaitEvent(comm, ev[i])          (what the executor folds onto its last launch)
//   write32    hipStreamWriteValue32(main, flag, i) + hipStreamWaitValue32(comm, flag, i, GEQ)
//   kflag      the work kernel's last workgroup stores i to a flag (agent-scope atomics); hipStreamWaitValue32(comm, flag, i, GEQ): NO packet on main
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/scratch/xstream_signal.hip -o /tmp/xstream_signal && /tmp/xstream_signal
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void work(float* buf, int iters, unsigned* count, unsigned* flag, unsigned value) {
    float v = buf[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    buf[blockIdx.x * 256 + threadIdx.x] = v;
    if (flag) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned prev = atomicAdd(count, 1u);
            if (prev % gridDim.x == gridDim.x - 1) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void tiny(unsigned* sink) { if (threadIdx.x == 0) sink[0] += 1; }

int main() {
    const int N = 2000, grid = 256;
    float* buf; unsigned *count, *sink, *flag;
    hipMalloc(&buf, grid * 256 * 4); hipMemset(buf, 0, grid * 256 * 4);
    hipMalloc(&count, 4); hipMemset(count, 0, 4);
    hipMalloc(&sink, 4); hipMemset(sink, 0, 4);
    if (hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) != hipSuccess) { printf("signal memory: failed\n"); return 1; }
    hipMemset(flag, 0, 8);
    hipStream_t main_s, comm;
    hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&comm, hipStreamNonBlocking, -1);
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    const int iters = 6000;
    double base = 0;
    const char* names[] = {"none", "event", "stop-event", "write32", "kflag"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(flag, 0, 8); hipMemset(count, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(t0, main_s);
            hipError_t err = hipSuccess;
            for (int i = 0; i < N && err == hipSuccess; ++i) {
                const unsigned val = (unsigned)(i + 1);
                if (mode == 2) hipExtLaunchKernelGGL(work, dim3(grid), dim3(256), 0, main_s, nullptr, ev[i], 0, buf, iters, count, (unsigned*)nullptr, val);
                else hipLaunchKernelGGL(work, dim3(grid), dim3(256), 0, main_s, buf, iters, count, mode == 4 ? flag : (unsigned*)nullptr, val);
                if (mode == 1) hipEventRecord(ev[i], main_s);
                if (mode == 1 || mode == 2) err = hipStreamWaitEvent(comm, ev[i], 0);
                if (mode == 3) { err = hipStreamWriteValue32(main_s, flag, val, 0); if (err == hipSuccess) err = hipStreamWaitValue32(comm, flag, val, hipStreamWaitValueGte, 0xFFFFFFFFu); }
                if (mode == 4) err = hipStreamWaitValue32(comm, flag, val, hipStreamWaitValueGte, 0xFFFFFFFFu);
                if (mode) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, comm, sink);
            }
            hipEventRecord(t1, main_s);
            hipEventSynchronize(t1);
            hipStreamSynchronize(comm);
            float ms; hipEventElapsedTime(&ms, t0, t1);
            if (err != hipSuccess) { printf("%-10s failed: %s\n", names[mode], hipGetErrorString(err)); break; }
            if (rep == 1) {
                const double us = ms * 1e3 / N;
                if (mode == 0) base = us;
                unsigned h = 0; hipMemcpy(&h, sink, 4, hipMemcpyDeviceToHost);
                printf("%-10s main stream %7.2f us per iteration  (+%.2f over no signal)   comm kernels run so far %u\n", names[mode], us, us - base, h);
            }
        }
    }
    return 0;
}
