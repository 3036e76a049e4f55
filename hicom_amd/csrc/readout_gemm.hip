// Readout MLP GEMM on matrix cores: y = act(x . w^T + b), x fp32 [M,K], w bf16 [N,K].
//
// Replaces build_mlp's nn.Linear / nn.GELU / nn.Linear (reference projector.py:307-312) applied
// to the window contexts (:559) -- the dense 2*M*(K*N) flop part of the compressor.
//
// Precision: the activation operand is an fp32 intermediate (attention context / GELU output);
// rounding it once to bf16 costs ~1e-3 max-abs, so it is split on the fly into
// bf16 hi + lo and contracted with two v_mfma_f32_16x16x32_bf16 into the same fp32 accumulator.
// Weights are exact bf16.
//
// Tiling: 64x64 output tile per 256-thread workgroup (2x2 waves, each 32x32 = 2x2 MFMA tiles).
// The K loop runs in stages of 64: operands are register-staged TWO stages ahead (global-load
// latency of ~1 us is several stage-times long at these small sizes; THREE stages in flight), converted/split while being
// written into a double-buffered LDS image with 144-byte rows (conflict-free ds_read_b128
// fragment reads), one workgroup barrier per stage.
// The store applies the packing row map of post_process_visual_feature (mm_utils.py:100-135).
#include <stdlib.h>

#include "common.hpp"

namespace hicom {

struct GemmParams {
    const float* x;
    const uint16_t* w;
    const void* b;
    int b_f32;
    int M, N, K, act;
    void* y;
    int y_f32;
    long ldy, row0;
    int nl_group;
};

constexpr int kRowB = 144;                 // bytes per 64-element bf16 row (128 B data + 16 B pad)
constexpr int kMatB = 64 * kRowB;          // one 64x64 bf16 operand image
constexpr int kBufB = 3 * kMatB;           // A_hi | A_lo | B

struct Stage {
    float4 a[4];     // 4 passes x (one 16-byte chunk of an A row): rows r, r+16, r+32, r+48
    u32x4 b[2];      // 2 passes x (one 16-byte chunk of a W row): rows r, r+32
};

// Global loads are issued so that consecutive lanes read consecutive 16-byte chunks of a row
// (a wave covers 4 full 256-B A-row segments / 8 full 128-B W-row segments per instruction).
__device__ __forceinline__ void stage_load(Stage& st, const float* const (&xa)[4], const uint16_t* const (&wb)[2], int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) st.a[i] = *reinterpret_cast<const float4*>(xa[i] + k0);
#pragma unroll
    for (int i = 0; i < 2; ++i) st.b[i] = *reinterpret_cast<const u32x4*>(wb[i] + k0);
}

__device__ __forceinline__ void stage_store(const Stage& st, char* buf, int tid) {
    const int ra = tid >> 4, ca = tid & 15;      // A: row within pass, 4-float chunk
    const int rb = tid >> 3, cb = tid & 7;       // B: row within pass, 8-bf16 chunk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v[4] = {st.a[i].x, st.a[i].y, st.a[i].z, st.a[i].w};
        u32x2 hi, lo;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint16_t h0, l0, h1, l1;
            split_bf16(v[2 * j], h0, l0);
            split_bf16(v[2 * j + 1], h1, l1);
            hi[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            lo[j] = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
        char* d = buf + (ra + 16 * i) * kRowB + 8 * ca;
        *reinterpret_cast<u32x2*>(d) = hi;
        *reinterpret_cast<u32x2*>(d + kMatB) = lo;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
        *reinterpret_cast<u32x4*>(buf + 2 * kMatB + (rb + 32 * i) * kRowB + 16 * cb) = st.b[i];
}

__global__ __launch_bounds__(256, 2) void readout_gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [2][kBufB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with
    // a private 4 MiB L2: all column tiles of one 64-row block of x run on ONE XCD, so that block
    // (295 KB fp32) is fetched from the Infinity Cache once per XCD pass and the weight matrix
    // (<= 2-25 MB) streams through; a plain (x,y) grid makes every XCD touch all of x AND w.
    const int nbx = (p.N + 63) >> 6, nby = (p.M + 63) >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int by = xcd + 8 * (slot / nbx), bx = slot - (slot / nbx) * nbx;
    if (by >= nby) return;
    const int m0 = by * 64, n0 = bx * 64;
    const int r16 = lane & 15, kg = lane >> 4;

    // staging assignment (see stage_load): clamp tail rows to a valid row, masked at the store
    const float* xa[4];
    const uint16_t* wb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int am = m0 + (tid >> 4) + 16 * i;
        am = am < p.M ? am : p.M - 1;
        xa[i] = p.x + (long)am * p.K + 4 * (tid & 15);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int bn = n0 + (tid >> 3) + 32 * i;
        bn = bn < p.N ? bn : p.N - 1;
        wb[i] = p.w + (long)bn * p.K + 8 * (tid & 7);
    }

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ns = p.K / 64;
    Stage st[3];
    stage_load(st[0], xa, wb, 0);
    stage_load(st[1], xa, wb, ns > 1 ? 64 : 0);
    stage_load(st[2], xa, wb, ns > 2 ? 128 : 64 * (ns - 1));
    stage_store(st[0], lds, tid);
    __syncthreads();

    auto compute = [&](const char* buf) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa_hi[2], fa_lo[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* ap = buf + (32 * wm + 16 * i + r16) * kRowB + 64 * kk + 16 * kg;
                fa_hi[i] = *reinterpret_cast<const bf16x8*>(ap);
                fa_lo[i] = *reinterpret_cast<const bf16x8*>(ap + kMatB);
                fb[i] = *reinterpret_cast<const bf16x8*>(buf + 2 * kMatB + (32 * wn + 16 * i + r16) * kRowB + 64 * kk + 16 * kg);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_hi[i], fb[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_lo[i], fb[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // Stage i lives in register set i % 3 while in flight and in LDS buffer i % 2 once staged.
    // Step S: issue the loads of stage S+3 (into the set stage S just vacated), compute stage S from
    // LDS, then convert + store stage S+1 (loaded two steps ago) into the other LDS buffer.
    // The loop body is deliberately BRANCH-FREE around the loads and stores (tail stages re-load
    // the last stage and store it where nobody reads): with conditional loads hipcc's waitcnt
    // pass loses count at the merge points and drains vmcnt(0) before every barrier, which
    // collapses the prefetch distance to zero.  The sched_barriers pin load / MFMA / store order.
    const int last_k = 64 * (ns - 1);
#define HICOM_GEMM_STEP(S, NXT, FREE)                                                     \
    do {                                                                                  \
        const int kl = 64 * ((S) + 3);                                                    \
        stage_load(st[FREE], xa, wb, kl < last_k ? kl : last_k);                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if ((S) < ns) compute(lds + ((S)&1) * kBufB);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                \
        stage_store(st[NXT], lds + (((S) + 1) & 1) * kBufB, tid);                         \
        __syncthreads();                                                                  \
    } while (0)
    for (int s = 0; s < ns; s += 3) {
        HICOM_GEMM_STEP(s, 1, 0);
        HICOM_GEMM_STEP(s + 1, 2, 1);
        HICOM_GEMM_STEP(s + 2, 0, 2);
    }
#undef HICOM_GEMM_STEP

    // epilogue: bias, activation, packed-row store.  C layout: col = lane & 15, rows 4*kg + j.
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 32 * wn + 16 * j + r16;
        if (n >= p.N) continue;
        float bias = 0.f;
        if (p.b) bias = p.b_f32 ? reinterpret_cast<const float*>(p.b)[n]
                                : bf16_to_f32(reinterpret_cast<const uint16_t*>(p.b)[n]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + 32 * wm + 16 * i + 4 * kg + q;
                if (m >= p.M) continue;
                float v = acc[i][j][q] + bias;
                if (p.act == HICOM_ACT_GELU) v = gelu_erf(v);
                const long orow = p.row0 + m + (p.nl_group > 0 ? m / p.nl_group : 0);
                if (p.y_f32) reinterpret_cast<float*>(p.y)[orow * p.ldy + n] = v;
                else reinterpret_cast<uint16_t*>(p.y)[orow * p.ldy + n] = f32_to_bf16(v);
            }
    }
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_readout_gemm_fwd(const float* x, const void* w, const void* b, int32_t b_dt,
                                      int32_t M, int32_t N, int32_t K, int32_t act,
                                      void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group,
                                      void* stream) {
    HICOM_REQUIRE(x && w && y, HICOM_EINVAL, "readout_gemm: NULL pointer");
    HICOM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, HICOM_EINVAL, "readout_gemm: bad shape M=%d N=%d K=%d (K %% 64)", M, N, K);
    HICOM_REQUIRE(ldy >= N && row0 >= 0 && nl_group >= 0, HICOM_EINVAL, "readout_gemm: bad output layout");
    HICOM_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0), HICOM_EINVAL, "readout_gemm: alignment");
    GemmParams p{x, (const uint16_t*)w, b, b_dt == HICOM_DT_F32, M, N, K, act, y, y_dt == HICOM_DT_F32,
                 (long)ldy, (long)row0, nl_group};
    const int nbx = (N + 63) / 64, nby = (M + 63) / 64;
    dim3 grid((unsigned)(8 * nbx * ((nby + 7) / 8)));
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(readout_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * kBufB);
        attr_set = true;
    }
    hipLaunchKernelGGL(readout_gemm_kernel, grid, dim3(256), 2 * kBufB, (hipStream_t)stream, p);
    return hicom_host::check_launch("readout_gemm");
}
