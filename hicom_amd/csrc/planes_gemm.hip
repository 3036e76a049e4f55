// Readout MLP GEMM on bf16 "planes": y = act((a_hi + a_lo) . w^T + b).
//
// fp32 intermediates that feed matrix cores travel between kernels as TWO bf16 matrices (hi = RNE
// bf16 of the value, lo = bf16 of the remainder; hi + lo carries 16 mantissa bits).  The producer
// (fused stream kernel / previous GEMM) writes the planes, so this GEMM has no conversion work at
// all: every operand tile goes HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), three stages of
// a ring in flight, counted s_waitcnt vmcnt + raw s_barrier (a __syncthreads would drain the DMA).
//
// Replaces nn.Linear / nn.GELU / nn.Linear of build_mlp on the window tokens
// (reference projector.py:307-312, :559).
//
// Tile: 64x64 per 256-thread workgroup (2x2 waves x 2x2 v_mfma_f32_16x16x32_bf16 tiles), BK = 64.
// LDS stage = A_hi | A_lo | W images of [64 rows][128 B] with the 16-byte chunk index XOR (row & 7)
// applied on the DMA *source* address and on the fragment read (the DMA destination is lane-linear),
// which leaves the ds_read_b128 fragment reads 2-way at worst.  XCD-aware tile order as in
// readout_gemm.hip.
#include "common.hpp"

namespace hicom {

struct PlanesGemmParams {
    const uint16_t* a_hi;
    const uint16_t* a_lo;
    const uint16_t* w;
    const void* b;
    int b_f32;
    int M, N, K, act;
    // output A: bf16 planes [M][N] (hidden activations for the next GEMM)
    uint16_t* o_hi;
    uint16_t* o_lo;
    // output B: packed rows of the final tensor (dtype y_f32 ? f32 : bf16)
    void* y;
    int y_f32;
    long ldy, row0;
    int nl_group;
};

constexpr int kPStage = 3 * 8192;     // bytes per ring stage: A_hi, A_lo, W images of 64 x 128 B
constexpr int kPRing = 3;

// HAS_LO = false: the activation is exactly bf16 (raw visual tokens feeding the k/v adaptor MLPs): no
// lo plane is fetched and half the MFMAs are issued.
template <bool HAS_LO>
__global__ __launch_bounds__(256, 2) void planes_gemm_kernel(PlanesGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [kPRing][kPStage]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kg = lane >> 4;
    const int nbx = (p.N + 63) >> 6, nby = (p.M + 63) >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int by = xcd + 8 * (slot / nbx), bx = slot - (slot / nbx) * nbx;
    if (by >= nby) return;
    const int m0 = by * 64, n0 = bx * 64;
    const int ns = p.K >> 6;

    // DMA assignment: 24 one-KiB pieces per stage (8 per operand image, 8 rows x 128 B each);
    // wave w issues pieces w, w+4, ..., i.e. 2 pieces of each operand.
    const int prow = lane >> 3, cpos = lane & 7;
    const uint16_t* src[6];
    int dst_off[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pi = wave + 4 * i;             // 0..23
        const int op = pi >> 3, piece = pi & 7;  // operand 0 A_hi, 1 A_lo, 2 W
        const int row = 8 * piece + prow;
        const int chunk = cpos ^ (row & 7);
        if (op < 2) {
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            src[i] = ((op == 0 || !HAS_LO) ? p.a_hi : p.a_lo) + (long)m * p.K + 8 * chunk;
        } else {
            int n = n0 + row;
            n = n < p.N ? n : p.N - 1;
            src[i] = p.w + (long)n * p.K + 8 * chunk;
        }
        dst_off[i] = op * 8192 + piece * 1024;
    }
    auto issue = [&](int s) {
        char* base = lds + (s % kPRing) * kPStage;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            // pieces 8..15 are the lo plane (i = 2, 3 of every wave: pi = wave + 4*i in [8, 16))
            if (!HAS_LO && (i == 2 || i == 3)) continue;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + 64 * s),
                                             (__attribute__((address_space(3))) void*)(base + dst_off[i]), 16, 0, 0);
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0);
    if (ns > 1) issue(1);

    for (int s = 0; s < ns; ++s) {
        // stage s has landed for every wave; stage s+1 may still be in flight
        if (s + 1 < ns) {
            if (HAS_LO) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // every wave is past compute(s-1): its ring slot is free for stage s+2
        if (s + 2 < ns) issue(s + 2);
        const char* st = lds + (s % kPRing) * kPStage;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa_hi[2], fa_lo[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ar = 32 * wm + 16 * i + r16;
                const int aoff = ar * 128 + 16 * ((4 * kk + kg) ^ (ar & 7));
                fa_hi[i] = *reinterpret_cast<const bf16x8*>(st + aoff);
                if (HAS_LO) fa_lo[i] = *reinterpret_cast<const bf16x8*>(st + 8192 + aoff);
                const int br = 32 * wn + 16 * i + r16;
                fb[i] = *reinterpret_cast<const bf16x8*>(st + 16384 + br * 128 + 16 * ((4 * kk + kg) ^ (br & 7)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_hi[i], fb[j], acc[i][j], 0, 0, 0);
                    if (HAS_LO) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_lo[i], fb[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue.  C layout: col = lane & 15, rows 4*kg + q.
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
                if (p.o_hi) {
                    uint16_t h, l;
                    split_bf16(v, h, l);
                    p.o_hi[(long)m * p.N + n] = h;
                    p.o_lo[(long)m * p.N + n] = l;
                }
                if (p.y) {
                    const long orow = p.row0 + m + (p.nl_group > 0 ? m / p.nl_group : 0);
                    if (p.y_f32) reinterpret_cast<float*>(p.y)[orow * p.ldy + n] = v;
                    else reinterpret_cast<uint16_t*>(p.y)[orow * p.ldy + n] = f32_to_bf16(v);
                }
            }
    }
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_planes_gemm_fwd(const void* a_hi, const void* a_lo, const void* w, const void* b, int32_t b_dt,
                                     int32_t M, int32_t N, int32_t K, int32_t act,
                                     void* out_hi, void* out_lo,
                                     void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group, void* stream) {
    HICOM_REQUIRE(a_hi && w, HICOM_EINVAL, "planes_gemm: NULL pointer");
    HICOM_REQUIRE((out_hi && out_lo) || y, HICOM_EINVAL, "planes_gemm: no output");
    HICOM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, HICOM_EINVAL, "planes_gemm: bad shape M=%d N=%d K=%d (K %% 64)", M, N, K);
    HICOM_REQUIRE(!y || (ldy >= N && row0 >= 0 && nl_group >= 0), HICOM_EINVAL, "planes_gemm: bad output layout");
    HICOM_REQUIRE(((uintptr_t)a_hi % 16 == 0) && ((uintptr_t)a_lo % 16 == 0) && ((uintptr_t)w % 16 == 0) && M < (1 << 30), HICOM_EINVAL,
                  "planes_gemm: alignment");
    PlanesGemmParams p{(const uint16_t*)a_hi, (const uint16_t*)a_lo, (const uint16_t*)w, b, b_dt == HICOM_DT_F32,
                       M, N, K, act, (uint16_t*)out_hi, (uint16_t*)out_lo, y, y_dt == HICOM_DT_F32, (long)ldy, (long)row0,
                       nl_group};
    const int nbx = (N + 63) / 64, nby = (M + 63) / 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(planes_gemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kPRing * kPStage);
        hipFuncSetAttribute(reinterpret_cast<const void*>(planes_gemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kPRing * kPStage);
        attr_set = true;
    }
    const dim3 grid((unsigned)(8 * nbx * ((nby + 7) / 8)));
    if (a_lo) hipLaunchKernelGGL(planes_gemm_kernel<true>, grid, dim3(256), kPRing * kPStage, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(planes_gemm_kernel<false>, grid, dim3(256), kPRing * kPStage, (hipStream_t)stream, p);
    return hicom_host::check_launch("planes_gemm");
}
