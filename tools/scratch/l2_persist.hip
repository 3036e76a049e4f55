// Dev probe (round 6): do lines a kernel pulled into an XCD's L2 survive the boundary to the NEXT kernel of the same stream?  (If so, CUs with
// spare miss-queue time in launch k can prefetch what few CUs of launch k+1 pull alone -- a CU takes ~30 GB/s from beyond L2, ~120 GB/s from L2.)
//   touch(region)   every workgroup reads the 512-KiB chunk of ITS XCD (hwreg XCC_ID) of `region`
//   read(region)    the same reads, timed per workgroup with s_memrealtime; reported: median GB/s per workgroup
// sequences:  cold:  flush, read(R)      warm-same-launch: read(R) twice inside one launch (second pass timed)
//             across: flush, touch(R), read(R)     across+traffic: flush, touch(R), stream(32 MB through every L2), read(R)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/scratch/l2_persist.hip -o /tmp/l2_persist && /tmp/l2_persist
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

constexpr unsigned long long kChunk = 512ull << 10;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}

// every thread: 16-byte loads striding the chunk, 8 in flight
__device__ __forceinline__ unsigned sweep(const char* chunk, unsigned long long bytes) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u acc = {0, 0, 0, 0};
    for (unsigned long long off = threadIdx.x * 16ull; off < bytes; off += 256 * 16 * 8) {
        v4u r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned long long o = off + u * 256ull * 16;
            o = o < bytes ? o : bytes - 16;
            r[u] = *reinterpret_cast<const v4u*>(chunk + o);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= r[u];
    }
    return acc.x ^ acc.y ^ acc.z ^ acc.w;
}

__global__ __launch_bounds__(256) void touch(const char* region, unsigned long long bytes, unsigned* sink) {
    const unsigned v = sweep(region + xcc_id() * kChunk, bytes);
    if (v == 0x12345678u) sink[0] = v;
}

// one 4-byte load per `stride` bytes of the chunk (is a line filled whole by a partial touch?)
__global__ __launch_bounds__(256) void touch_sparse(const char* region, unsigned long long bytes, unsigned stride, unsigned* sink) {
    const char* chunk = region + xcc_id() * kChunk;
    unsigned acc = 0;
    for (unsigned long long off = (unsigned long long)threadIdx.x * stride; off < bytes; off += 256ull * stride) acc ^= *reinterpret_cast<const unsigned*>(chunk + off);
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void read_timed(const char* region, unsigned long long bytes, int passes, unsigned* sink, unsigned* ticks) {
    const char* chunk = region + xcc_id() * kChunk;
    unsigned v = 0;
    for (int p = 0; p + 1 < passes; ++p) v ^= sweep(chunk, bytes);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    v ^= sweep(chunk, bytes);
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = (unsigned)(t1 - t0);
    if (v == 0x12345678u) sink[0] = v;
}

__global__ __launch_bounds__(256) void stream(const char* buf, unsigned long long bytes_per_wg, unsigned* sink) {
    const unsigned v = sweep(buf + blockIdx.x * bytes_per_wg, bytes_per_wg);
    if (v == 0x12345678u) sink[0] = v;
}

int main() {
    char *big, *reg;
    unsigned *sink, *ticks;
    const unsigned long long kBig = 1ull << 30;
    hipMalloc(&big, kBig); hipMemset(big, 1, kBig);
    hipMalloc(&reg, 64ull << 20); hipMemset(reg, 2, 64ull << 20);
    hipMalloc(&sink, 64); hipMalloc(&ticks, 4096);
    auto flush = [&] { hipLaunchKernelGGL(stream, dim3(1024), dim3(256), 0, 0, big, kBig / 1024, sink); };       // 1 GiB through every cache
    auto report = [&](const char* name, unsigned long long bytes) {
        hipDeviceSynchronize();
        std::vector<unsigned> h(256);
        hipMemcpy(h.data(), ticks, 1024, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        auto gbs = [&](unsigned t) { return bytes / (t * 10e-9) / 1e9; };
        printf("%-58s %4llu KiB per workgroup: median %6.1f GB/s per workgroup  (p10 %6.1f  p90 %6.1f)   median %5.2f us\n", name, bytes >> 10, gbs(h[128]), gbs(h[230]), gbs(h[25]),
               h[128] / 100.0);
    };
    for (unsigned long long bytes : {64ull << 10, 512ull << 10}) {
        for (int rep = 0; rep < 2; ++rep) {
            const char* r = reg + (rep * 2 + (bytes > (64ull << 10))) * 8 * kChunk;
            flush();
            hipLaunchKernelGGL(read_timed, dim3(256), dim3(256), 0, 0, r, bytes, 1, sink, ticks);
            report("cold (flushed; from HBM)", bytes);
            hipLaunchKernelGGL(read_timed, dim3(256), dim3(256), 0, 0, r, bytes, 2, sink, ticks);
            report("second pass inside one launch (the XCD's L2)", bytes);
            flush();
            hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, 0, r, bytes, sink);
            hipLaunchKernelGGL(read_timed, dim3(256), dim3(256), 0, 0, r, bytes, 1, sink, ticks);
            report("touched by the PREVIOUS launch (across the boundary)", bytes);
            flush();
            hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, 0, r, bytes, sink);
            hipLaunchKernelGGL(stream, dim3(256), dim3(256), 0, 0, big, (16ull << 20) / 256, sink);              // 16 MB = 2 MB through each L2
            hipLaunchKernelGGL(read_timed, dim3(256), dim3(256), 0, 0, r, bytes, 1, sink, ticks);
            report("touched two launches ago, 2 MB per L2 streamed in between", bytes);
            for (unsigned stride : {128u, 64u, 32u}) {
                flush();
                hipLaunchKernelGGL(touch_sparse, dim3(256), dim3(256), 0, 0, r, bytes, stride, sink);
                hipLaunchKernelGGL(read_timed, dim3(256), dim3(256), 0, 0, r, bytes, 1, sink, ticks);
                char nm[96];
                snprintf(nm, sizeof nm, "previous launch touched 4 B per %u B", stride);
                report(nm, bytes);
            }
            flush();
            hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, 0, r, bytes, sink);
            hipLaunchKernelGGL(stream, dim3(1024), dim3(256), 0, 0, big, (256ull << 20) / 1024, sink);          // 256 MB: beyond every L2
            hipLaunchKernelGGL(read_timed, dim3(256), dim3(256), 0, 0, r, bytes, 1, sink, ticks);
            report("touched, then 256 MB streamed (Infinity Cache at best)", bytes);
        }
    }
    return 0;
}
