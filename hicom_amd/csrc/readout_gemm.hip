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
// Tiling: 64x64 output tile per 256-thread workgroup (2x2 waves, each 32x32 = 2x2 MFMA tiles),
// BK = 32; operands are register-staged (next k-step's global loads fly under the MFMAs) into
// 80-byte-stride LDS rows, which makes every ds_read_b128 fragment read conflict-free.
// The store applies the packing row map of post_process_visual_feature (mm_utils.py:100-135).
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

constexpr int kLdsRow = 80;   // bytes per 32-element bf16 row (64 B data + 16 B pad)

__global__ __launch_bounds__(256) void readout_gemm_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char lds[3 * 64 * kLdsRow];
    char* Ahi = lds;
    char* Alo = lds + 64 * kLdsRow;
    char* Bs = lds + 2 * 64 * kLdsRow;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int r16 = lane & 15, kg = lane >> 4;

    // staging assignment: thread -> (row, 8-element k chunk)
    const int srow = tid >> 2, skc = tid & 3;
    int am = m0 + srow; am = am < p.M ? am : p.M - 1;
    int bn = n0 + srow; bn = bn < p.N ? bn : p.N - 1;
    const float* xa = p.x + (long)am * p.K + 8 * skc;
    const uint16_t* wb = p.w + (long)bn * p.K + 8 * skc;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 a0 = *reinterpret_cast<const float4*>(xa);
    float4 a1 = *reinterpret_cast<const float4*>(xa + 4);
    u32x4 bw = *reinterpret_cast<const u32x4*>(wb);

    const int nk = p.K / 32;
    for (int ks = 0; ks < nk; ++ks) {
        __syncthreads();
        {
            const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            u32x4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint16_t h0, l0, h1, l1;
                split_bf16(v[2 * i], h0, l0);
                split_bf16(v[2 * i + 1], h1, l1);
                hi[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
                lo[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
            }
            *reinterpret_cast<u32x4*>(Ahi + srow * kLdsRow + 16 * skc) = hi;
            *reinterpret_cast<u32x4*>(Alo + srow * kLdsRow + 16 * skc) = lo;
            *reinterpret_cast<u32x4*>(Bs + srow * kLdsRow + 16 * skc) = bw;
        }
        __syncthreads();
        if (ks + 1 < nk) {
            a0 = *reinterpret_cast<const float4*>(xa + 32 * (ks + 1));
            a1 = *reinterpret_cast<const float4*>(xa + 32 * (ks + 1) + 4);
            bw = *reinterpret_cast<const u32x4*>(wb + 32 * (ks + 1));
        }
        bf16x8 fa_hi[2], fa_lo[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ar = 32 * wm + 16 * i + r16;
            fa_hi[i] = *reinterpret_cast<const bf16x8*>(Ahi + ar * kLdsRow + 16 * kg);
            fa_lo[i] = *reinterpret_cast<const bf16x8*>(Alo + ar * kLdsRow + 16 * kg);
            const int br = 32 * wn + 16 * i + r16;
            fb[i] = *reinterpret_cast<const bf16x8*>(Bs + br * kLdsRow + 16 * kg);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_hi[i], fb[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_lo[i], fb[j], acc[i][j], 0, 0, 0);
            }
    }

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
    HICOM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0, HICOM_EINVAL, "readout_gemm: bad shape M=%d N=%d K=%d (K %% 32)", M, N, K);
    HICOM_REQUIRE(ldy >= N && row0 >= 0 && nl_group >= 0, HICOM_EINVAL, "readout_gemm: bad output layout");
    HICOM_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0), HICOM_EINVAL, "readout_gemm: alignment");
    GemmParams p{x, (const uint16_t*)w, b, b_dt == HICOM_DT_F32, M, N, K, act, y, y_dt == HICOM_DT_F32,
                 (long)ldy, (long)row0, nl_group};
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    hipLaunchKernelGGL(readout_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hicom_host::check_launch("readout_gemm");
}
