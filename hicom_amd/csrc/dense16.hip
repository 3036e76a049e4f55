// Dense 16-bit MFMA GEMM over ALL tokens:  C[M,N] = epilogue(A[M,K] . W[N,K]^T + b)   (M = T*729 ~ 46 656 rows)
//
// The compressor's neighbours and adaptors that are genuinely matrix-core bound:
//   * the SigLIP pooling-head projection in front of the key stream, frames_embed = x + head.mlp(head.layernorm(x))
//     (reference encoder.py:284-286; HF SiglipMLP: fc1 -> gelu_pytorch_tanh -> fc2): 925 GFLOP at 64 frames;
//   * the k / v adaptor MLPs of `local43_adaptkv_global32` over all tokens (reference projector.py:533-534);
//   * the key norms ||W_k x + b_k|| of the clip-scale global stage (reference projector.py:184-186).
// Both operands are K-contiguous (nn.Linear keeps W as [N,K]): an NT GEMM, fragments read along K by ds_read_b128.
//
// Structure: 128 x 128 output tile per 256-thread workgroup (2 x 2 waves of 64 x 64 = 4 x 4 MFMA blocks), BK = 64, both
// operand tiles HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) into ONE 32-KB stage with the chunk ^ (row >> 1) swizzle on the
// DMA source and the fragment read, two barriers per K step; three to four workgroups are resident per CU (<= 168 VGPR,
// 32 KB of LDS each) and it is the interleaving of THEIR phases that overlaps staging with MFMA work
// (cdna_hip_programming.md §5 "Measured: the optimization ladder": the 128^2 two-barrier structure, ~0.9 PF on that
// guide's box; the 256^2 8-phase schedule is the next step up and is not built here).  XCD-aware tile order: one
// XCD walks a contiguous run of tiles that share A rows.  Product computed transposed (W fragment as the MFMA A
// operand): a lane holds 4 consecutive output columns of one row; the epilogue regroups them through LDS into 8 consecutive
// columns per lane so that stores, bias and residual accesses are whole 128-byte lines (dense_epilogue_rows).
// What was measured on the way (head projection, 46 656 x 4 352 x 1 152, fp16; DESIGN.md §3.4 has the table): the 8-byte
// MFMA-layout stores were 185-210 us of an 800-us launch; a persistent 256 x 256 x 32 four-stage LDS-DMA ring (one
// workgroup per CU) ran its K loop at 0.9-1.0 PFLOP/s but lost it again in an epilogue nothing overlaps (tanh-GELU is ~25
// VALU ops per output against 2 304 MFMA flops at K = 1 152) and in the ~130 clocks per 1-KB DMA piece that the issuing
// COMPUTE waves queue behind the CU's address path -- it ended 15 % slower than this kernel and is not kept.
// fp16 operands for normalised activations (11 significand bits; bf16 weights convert exactly), bf16 operands when A is
// the raw bf16 token stream.
#include <stdlib.h>

#include "common.hpp"

namespace hicom {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define HICOM_ACT_GELU_TANH 2

struct DenseParams {
    const uint16_t* a;      // [M, lda] 16-bit (fp16 or bf16 bit patterns)
    const uint16_t* w;      // [N, ldw]
    long lda, ldw;
    const void* b;          // bias [N] (bf16 | f32) or NULL
    int b_f32;
    int M, N, K, act;
    // outputs (any subset)
    _Float16* o16;          // fp16 [M, ldo]; columns [N, n_store) are written as zeros (K padding of the next GEMM)
    long ldo;
    int n_store;
    _Float16* pre16;        // fp16 [M, ldpre]: acc + b BEFORE the activation (what the activation's backward needs), or NULL; row epilogue only
    long ldpre;
    void* y;                // bf16 | f32 [M, ldy]: value + residual
    int y_f32;
    long ldy;
    const uint16_t* res;    // bf16 [M, ldr] residual added before the y store (or NULL)
    long ldr;
    float* ssq;             // [ceil(N / 64)][M] partial row sums of squares of (acc + b), one row per 64-column slice (or NULL)
    // row-dot output (row-contiguous epilogue only): rdot[(n / 64) * M + m] = sum over the 64-column slice of dotv[n] * (value + res)[m, n]
    const void* dotv;       // [N] bf16 | f32
    int dotv_f32;
    float* rdot;            // [ceil(N / 64)][M] (or NULL)
    int tiles_m, tiles_n;
    // optional per-row additive term from three table rows (the projected positional embedding W . pos of token m):
    //   + tab[t0 + m / (H*W)][n] + tab[y0 + (m / W) % H][n] + tab[x0 + m % W][n],   tab f32 [*, tab_ld]
    const float* tab;
    long tab_ld;
    int H, W, t0, y0, x0;
    // optional SECOND problem of the same shape in the same launch (hicom_dense16_gemm_pair_fwd): blocks >= tiles_m * tiles_n run it.
    // Its operands, bias and fp16 output replace a / w / b / o16 / pre16; every other field is shared.
    const uint16_t* a2;
    const uint16_t* w2;
    const void* b2;
    _Float16* o16_2;
    _Float16* pre16_2;
};

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh u) = x / (1 + e^(-2u)),  u = 0.79788456 (x + 0.044715 x^3):  -2u log2(e) = x (c0 + c1 x^2)
    const float t = x * x;
    const float arg = x * fmaf(t, -0.10294324f, -2.3022082f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
}

// Epilogue of one wave's [16 MI rows] x [16 NJ columns] block at (mw, nw): the lane holds columns n .. n+3 (n = nw + 16 j + 4 kg)
// of row m = mw + 16 i + r16.  bias, positional-table term, activation, fp16 / packed stores, row sums of squares per
// 64-column slice.
template <int MI, int NJ>
__device__ __forceinline__ void dense_epilogue(const DenseParams& p, f32x4 (&acc)[NJ][MI], int mw, int nw, int r16, int kg) {
    float rss[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rss[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nw + 16 * j + 4 * kg;
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.b) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int nn = n + qq < p.N ? n + qq : p.N - 1;
                bias[qq] = p.b_f32 ? reinterpret_cast<const float*>(p.b)[nn] : bf16_to_f32(reinterpret_cast<const uint16_t*>(p.b)[nn]);
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + 16 * i + r16;
            float v[4];
            if (p.tab && n < p.N) {
                const int mm = m < p.M ? m : p.M - 1;
                const int t = mm / (p.H * p.W), rem2 = mm - t * (p.H * p.W), yy = rem2 / p.W, xx = rem2 - yy * p.W;
                const float4 a0 = *reinterpret_cast<const float4*>(p.tab + (long)(p.t0 + t) * p.tab_ld + n);
                const float4 a1 = *reinterpret_cast<const float4*>(p.tab + (long)(p.y0 + yy) * p.tab_ld + n);
                const float4 a2 = *reinterpret_cast<const float4*>(p.tab + (long)(p.x0 + xx) * p.tab_ld + n);
                acc[j][i][0] += a0.x + a1.x + a2.x; acc[j][i][1] += a0.y + a1.y + a2.y;
                acc[j][i][2] += a0.z + a1.z + a2.z; acc[j][i][3] += a0.w + a1.w + a2.w;
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                v[qq] = (n + qq < p.N) ? acc[j][i][qq] + bias[qq] : 0.f;
                if (p.ssq) rss[i] = fmaf(v[qq], v[qq], rss[i]);
                if (p.act == HICOM_ACT_GELU) v[qq] = gelu_erf(v[qq]);
                else if (p.act == HICOM_ACT_GELU_TANH) v[qq] = gelu_tanh(v[qq]);
            }
            if (m >= p.M) continue;
            if (p.o16 && n < p.n_store) {
                half4 hv;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) hv[qq] = (_Float16)fminf(fmaxf((n + qq < p.N) ? v[qq] : 0.f, -65504.f), 65504.f);
                *reinterpret_cast<half4*>(p.o16 + (long)m * p.ldo + n) = hv;       // (ldo, n_store multiples of 4: host-checked)
            }
            if (p.y && n < p.N) {
                if (p.res) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(p.res + (long)m * p.ldr + n);
                    v[0] += bf16lo_to_f32(rr.x); v[1] += bf16hi_to_f32(rr.x); v[2] += bf16lo_to_f32(rr.y); v[3] += bf16hi_to_f32(rr.y);
                }
                if (p.y_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (long)m * p.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + (long)m * p.ldy + n) =
                        make_uint2(f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16), f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
            }
        }
        if (p.ssq && (j & 3) == 3) {
            // sum over the 64-column slice (nw + 16 (j - 3)) / 64 of the row: its 4 k-groups live in lanes r16 + 16 kg
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                float sacc = rss[i];
                sacc += __shfl_xor(sacc, 16, 64);
                sacc += __shfl_xor(sacc, 32, 64);
                const int m = mw + 16 * i + r16, nb = nw + 16 * (j - 3);
                if (kg == 0 && m < p.M && nb < p.N) p.ssq[(long)(nb >> 6) * p.M + m] = sacc;
                rss[i] = 0.f;
            }
        }
    }
}

// Row-contiguous epilogue through a wave-private 4-KB LDS staging block.  The MFMA layout gives a lane 4 columns of one row
// (8-byte fp16 stores, 32 contiguous bytes per row and instruction: measured 185-210 us of an 800-us launch at the head
// projection's shapes, tools/gpu_dense_ring.sh).  Here each [16 rows x 64 columns] fp32 block goes to LDS (ds_write_b128,
// chunk c of row r at c ^ r: the 8 lanes of a store group hit 8 distinct bank quads) and comes back with 8 consecutive
// columns of one row per lane (rows 8h + lane / 8, two ds_read_b128 at chunks (2q, 2q+1) ^ row: conflict-free in the
// b128 lane groups), so that bias / residual / table loads are 16- or 32-byte reads and every store instruction writes
// eight whole 128-byte lines (16-bit outputs) or 256 contiguous bytes per row (fp32).  Same-wave LDS operations execute in
// issue order: no barrier between the staging writes and reads.
// bias_staged: the wave's bias slice (columns nw .. nw + 16 NJ, raw fp32 / bf16) already lies at the head of `stage` (the ring
// kernel's loaders put it there by LDS-DMA a few K steps earlier): read before the first staging write, no global latency.
template <int MI, int NJ, bool PRE = false>
__device__ __forceinline__ void dense_epilogue_rows(const DenseParams& p, f32x4 (&acc)[NJ][MI], int mw, int nw, int lane, float* stage,
                                                    bool bias_staged = false) {
    static_assert(NJ % 4 == 0, "64-column slices");
    const int r16 = lane & 15, kg = lane >> 4;
    const int q = lane & 7, rr = lane >> 3;
    float sbias[NJ / 4][8];
    if (bias_staged) {
#pragma unroll
        for (int jq = 0; jq < NJ / 4; ++jq) {
            if (p.b_f32) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(stage + 64 * jq + 8 * q);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(stage + 64 * jq + 8 * q + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sbias[jq][e] = b0[e]; sbias[jq][4 + e] = b1[e]; }
            } else {
                const u32x4 g = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(stage) + 64 * jq + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sbias[jq][2 * e] = bf16lo_to_f32(g[e]); sbias[jq][2 * e + 1] = bf16hi_to_f32(g[e]); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int jq = 0; jq < NJ / 4; ++jq) {
        const int n = nw + 64 * jq + 8 * q;                // this lane's 8 columns after the regroup
        const bool n_ok = n < p.N;                          // N % 8 == 0 (host-checked): a chunk is inside or outside as a whole
        float bias[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = 0.f;
        if (bias_staged) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bias[e] = sbias[jq][e];
        } else if (p.b && n_ok) {
            if (p.b_f32) {
                const float4 b0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.b) + n);
                const float4 b1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.b) + n + 4);
                bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
            } else {
                const u32x4 g = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.b) + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bias[2 * e] = bf16lo_to_f32(g[e]); bias[2 * e + 1] = bf16hi_to_f32(g[e]); }
            }
        }
        float dv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dv[e] = 0.f;
        if (p.rdot && n_ok) {
            if (p.dotv_f32) {
                const float4 d0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.dotv) + n);
                const float4 d1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.dotv) + n + 4);
                dv[0] = d0.x; dv[1] = d0.y; dv[2] = d0.z; dv[3] = d0.w; dv[4] = d1.x; dv[5] = d1.y; dv[6] = d1.z; dv[7] = d1.w;
            } else {
                const u32x4 g = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.dotv) + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { dv[2 * e] = bf16lo_to_f32(g[e]); dv[2 * e + 1] = bf16hi_to_f32(g[e]); }
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                *reinterpret_cast<f32x4*>(stage + r16 * 64 + 4 * ((4 * jj + kg) ^ r16)) = acc[4 * jq + jj][i];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 8 * h + rr, m = mw + 16 * i + row;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + 4 * ((2 * q) ^ row));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * 64 + 4 * ((2 * q + 1) ^ row));
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const int mm = m < p.M ? m : p.M - 1;
                if (p.tab && n_ok) {
                    const int t = mm / (p.H * p.W), rem2 = mm - t * (p.H * p.W), yy = rem2 / p.W, xx = rem2 - yy * p.W;
                    const float* t0 = p.tab + (long)(p.t0 + t) * p.tab_ld + n;
                    const float* t1 = p.tab + (long)(p.y0 + yy) * p.tab_ld + n;
                    const float* t2 = p.tab + (long)(p.x0 + xx) * p.tab_ld + n;
#pragma unroll
                    for (int e4 = 0; e4 < 2; ++e4) {
                        const float4 a0 = *reinterpret_cast<const float4*>(t0 + 4 * e4);
                        const float4 a1 = *reinterpret_cast<const float4*>(t1 + 4 * e4);
                        const float4 a2 = *reinterpret_cast<const float4*>(t2 + 4 * e4);
                        v[4 * e4 + 0] += a0.x + a1.x + a2.x; v[4 * e4 + 1] += a0.y + a1.y + a2.y;
                        v[4 * e4 + 2] += a0.z + a1.z + a2.z; v[4 * e4 + 3] += a0.w + a1.w + a2.w;
                    }
                }
                float rss = 0.f;
                if constexpr (PRE) {
                    // the value BEFORE the activation as a second fp16 output (own instantiation: the plain kernel keeps its 112 VGPRs)
                    if (m < p.M && n_ok) {
                        half8 hv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) hv[e] = (_Float16)fminf(fmaxf(v[e] + bias[e], -65504.f), 65504.f);
                        *reinterpret_cast<half8*>(p.pre16 + (long)m * p.ldpre + n) = hv;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = n_ok ? v[e] + bias[e] : 0.f;
                    rss = fmaf(v[e], v[e], rss);
                    if (p.act == HICOM_ACT_GELU) v[e] = gelu_erf(v[e]);
                    else if (p.act == HICOM_ACT_GELU_TANH) v[e] = gelu_tanh(v[e]);
                }
                if (p.ssq) {
                    // the 8 lanes of a row hold its 64-column slice
                    rss += __shfl_xor(rss, 1, 64);
                    rss += __shfl_xor(rss, 2, 64);
                    rss += __shfl_xor(rss, 4, 64);
                    const int nb = nw + 64 * jq;
                    if (q == 0 && m < p.M && nb < p.N) p.ssq[(long)(nb >> 6) * p.M + m] = rss;
                }
                if (m < p.M && p.o16 && n < p.n_store) {
                    half8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (_Float16)fminf(fmaxf(v[e], -65504.f), 65504.f);
                    *reinterpret_cast<half8*>(p.o16 + (long)m * p.ldo + n) = hv;
                }
                if (!p.y && !p.rdot) continue;
                if (p.res && n_ok) {
                    const u32x4 g = *reinterpret_cast<const u32x4*>(p.res + (long)mm * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] += bf16lo_to_f32(g[e]); v[2 * e + 1] += bf16hi_to_f32(g[e]); }
                }
                if (p.rdot) {
                    // the 8 lanes of a row hold its 64-column slice (dv is zero outside N)
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) d = fmaf(dv[e], v[e], d);
                    d += __shfl_xor(d, 1, 64);
                    d += __shfl_xor(d, 2, 64);
                    d += __shfl_xor(d, 4, 64);
                    const int nb = nw + 64 * jq;
                    if (q == 0 && m < p.M && nb < p.N) p.rdot[(long)(nb >> 6) * p.M + m] = d;
                }
                if (p.y && n_ok && m < p.M) {
                    if (p.y_f32) {
                        float* yp = reinterpret_cast<float*>(p.y) + (long)m * p.ldy + n;
                        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        u32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.y) + (long)m * p.ldy + n) = o;
                    }
                }
            }
        }
    }
}

// 128 x 128 tile, 2 x 2 waves of 64 x 64, ONE 32-KB stage, two barriers per K step; the workgroups resident on a CU (four at
// 112 VGPRs with the row epilogue) overlap each other's staging, MFMA and epilogue phases.
template <bool BF16, bool ROWS, bool PRE = false>
__global__ __launch_bounds__(256, 3) void dense16_gemm_kernel(DenseParams p) {
    constexpr int NWAVE = 4, WC = 2, MI = 4, TM = 128, TN = 128;
    constexpr int PPW = (TM + TN) / 8 / NWAVE;                  // one-KiB DMA pieces (8 rows x 128 B) per wave and stage
    constexpr int PA = TM / 8 / NWAVE;                          // ... of which the first PA are A pieces
    extern __shared__ __attribute__((aligned(16))) char lds[];  // A image [TM][128 B] | W image [TN][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave - wr * WC;
    const int r16 = lane & 15, kg = lane >> 4;
    // XCD-aware bijective remap (cdna_hip_programming.md §5 "XCD swizzle must be bijective")
    const int nwg = p.tiles_m * p.tiles_n;
    int orig = blockIdx.x;
    if (orig >= nwg) {                      // second problem of a paired launch (uniform per workgroup: scalar moves)
        orig -= nwg;
        p.a = p.a2; p.w = p.w2; p.b = p.b2; p.o16 = p.o16_2; p.pre16 = p.pre16_2;
    }
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // super-tile order inside the run: blocks of RM row tiles, column-major inside a block, so that the tiles an XCD has in
    // flight share RM A panels and a few W panels in ITS L2 instead of streaming all of W once per row tile
    constexpr int RM = 8;
    const int per_blk = RM * p.tiles_n, blk = wg / per_blk, rem = wg - blk * per_blk;
    const int rows_in = min(RM, p.tiles_m - blk * RM);
    const int bn = rem / rows_in, bm = blk * RM + (rem - bn * rows_in);
    const int m0 = bm * TM, n0 = bn * TN;

    const int prow = lane >> 3, cpos = lane & 7;
    unsigned soff[PPW];                                 // element offsets (< 2^32: the largest operand here is 406 MB)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        if (i < PA) {
            const int row = 8 * (wave + NWAVE * i) + prow;
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            soff[i] = (unsigned)((long)m * p.lda + 8 * (cpos ^ ((row >> 1) & 7)));
        } else {
            const int row = 8 * (wave + NWAVE * (i - PA)) + prow;
            int n = n0 + row;
            n = n < p.N ? n : p.N - 1;
            soff[i] = (unsigned)((long)n * p.ldw + 8 * (cpos ^ ((row >> 1) & 7)));
        }
    }
    const int swz = (r16 >> 1) & 7;
    const char* a_rd = lds + (16 * MI * wr + r16) * 128;
    const char* w_rd = lds + TM * 128 + (64 * wc + r16) * 128;

    f32x4 acc[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ns = p.K >> 6;
    for (int s = 0; s < ns; ++s) {
        lds_barrier();                                        // every wave is done reading the previous stage
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const uint16_t* src = (i < PA ? p.a : p.w) + soff[i] + 64 * s;
            const int piece = i < PA ? wave + NWAVE * i : TM / 8 + wave + NWAVE * (i - PA);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                             (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = 16 * ((4 * ks + kg) ^ swz);
            if constexpr (BF16) {
                bf16x8 wf[4], af[MI];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(w_rd + i * 2048 + off);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_rd + i * 2048 + off);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
            } else {
                half8 wf[4], af[MI];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const half8*>(w_rd + i * 2048 + off);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const half8*>(a_rd + i * 2048 + off);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[j][i], 0, 0, 0);
            }
        }
    }

    if constexpr (!ROWS) {
        dense_epilogue<MI, 4>(p, acc, m0 + 16 * MI * wr, n0 + 64 * wc, r16, kg);
    } else {
        lds_barrier();                                        // every wave is done reading the last stage: the buffer becomes staging space
        dense_epilogue_rows<MI, 4, PRE>(p, acc, m0 + 16 * MI * wr, n0 + 64 * wc, lane, reinterpret_cast<float*>(lds) + wave * 1024);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// TN form: C[M, N] = A^T B,  A [Kt, lda] (element (k, m)), B [Kt, ldb] (element (k, n)), contraction over the Kt ROWS.
// The weight gradients of the token-stream layers, dW = dY^T X over all T*729 tokens (training path: the k / v adaptor MLPs,
// the SigLIP head projection; reference train.py:700-738 through autograd): both operands arrive token-major, i.e. with the
// contraction index as the SLOW axis, so the K-contiguous fragment reads of the NT kernel above do not apply.  Same skeleton
// (128 x 128 tile, 2 x 2 waves of 64 x 64, one 32-KB stage of 64 token rows, LDS-DMA, two barriers per stage, 3-4 workgroups per
// CU), with the fragments fetched by ds_read_b64_tr_b16 (cdna_hip_programming.md T10): a 16-lane group reads a 4-row x
// 16-column block and each lane receives ITS column's four row values -- the MFMA operand layout with k = token.  Two such
// reads (rows kb + 4g + e and kb + 16 + 4g + e) fill the 8 k-slots of a 16x16x32 operand; both operands use the same slot -> token
// map, so the contraction is unaffected by it.  Image: plain 256-byte rows with the chunk XOR ((row & 3) << 2) | ((row >> 2) & 3)
// of T10 (b), applied on the DMA source side.
// The output is small (E x E) and the contraction long: the token axis is SPLIT over blockIdx.y and every split writes its own
// f32 partial tile (deterministic; summed by hicom_partials_sum_fwd), which also gives the launch enough workgroups.
struct DenseTnParams {
    const uint16_t* a;
    const uint16_t* b;
    long lda, ldb;
    int M, N, Kt;
    float* c;               // [splits][M][ldc]
    long ldc;
    int tiles_m, tiles_n, splits;
};

__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <bool BF16>
__global__ __launch_bounds__(256, 3) void dense16_tn_kernel(DenseTnParams p) {
    constexpr int NWAVE = 4, MI = 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // A image [64 rows][256 B] | B image [64 rows][256 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, kg = lane >> 4;
    // Block -> (tile, split), XCD-aware (speed only): block b runs on XCD b % 8, and with a split count that is a multiple of 8 every
    // XCD works on ITS slices of the token axis only -- the tiles of a slice walk the same token rows at about the same time, so a
    // slice's operand panels stream through one L2 once instead of through all eight (measured: 389 -> see DESIGN.md §3.4)
    const int tiles = p.tiles_m * p.tiles_n;
    int tile, split;
    if ((p.splits & 7) == 0) {
        const int idx = blockIdx.x >> 3;
        split = (blockIdx.x & 7) + 8 * (idx / tiles);
        tile = idx - (idx / tiles) * tiles;
    } else {
        split = blockIdx.x / tiles;
        tile = blockIdx.x - split * tiles;
    }
    const int bm = tile / p.tiles_n, bn = tile - bm * p.tiles_n;
    const int m0 = bm * 128, n0 = bn * 128;
    const int ns_total = (p.Kt + 63) >> 6;
    const int per = (ns_total + p.splits - 1) / p.splits;
    const int s0 = split * per, s1 = min(ns_total, s0 + per);

    // DMA: a piece = 4 token rows x 256 B; 16 A pieces then 16 B pieces per stage; wave w issues pieces w, w + 4, ... of each
    const int prow = lane >> 4, pos = lane & 15;
    long col_off[8];                                     // element offset of this lane's 16-byte chunk inside its source row
    int lrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool isa = i < 4;
        const int r = 4 * (wave + NWAVE * (i & 3)) + prow;
        int col = (isa ? m0 : n0) + 8 * (pos ^ tn_swz(r));
        const int lim = (isa ? p.M : p.N) - 8;           // (columns past the matrix: any valid chunk -- those outputs are not stored)
        col = col < lim ? col : (lim > 0 ? lim : 0);
        col_off[i] = col;
        lrow[i] = r;
    }
    // transposed fragment reads: group g = kg reads rows kb + 4 sig(g) + q, q = (lane >> 2) & 3, columns 4 pp .. 4 pp + 3 of the 16-column
    // block; sig = (0, 2, 1, 3): the two blocks of a 32-lane half sit 8 rows apart (conflict-free on this image, T10), and since
    // both operands use the same group -> rows map the contraction does not see it
    const int q = (lane >> 2) & 3, pp = lane & 3;
    const int sg = ((kg & 1) << 1) | (kg >> 1);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(lds);
    f32x4 acc[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int s = s0; s < s1; ++s) {
        lds_barrier();                                        // every wave is done reading the previous stage
        const int valid = p.Kt - 64 * s;                      // token rows of this stage (>= 64 except in the last one)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool isa = i < 4;
            int kk = 64 * s + lrow[i];
            kk = kk < p.Kt ? kk : p.Kt - 1;
            const uint16_t* src = (isa ? p.a + (long)kk * p.lda : p.b + (long)kk * p.ldb) + col_off[i];
            const int piece = (isa ? 0 : 16) + wave + NWAVE * (i & 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                             (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (valid < 64) {                                     // ragged end of the token axis: rows past it contribute zeros
            for (int c = tid; c < (64 - valid) * 32; c += 256) {
                const int rr = valid + (c >> 5), w16 = c & 31;          // 32 16-byte chunks per row pair (A row | B row)
                *reinterpret_cast<u32x4*>(lds + (w16 < 16 ? 0 : 16384) + rr * 256 + 16 * (w16 & 15)) = u32x4{0, 0, 0, 0};
            }
            lds_barrier();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // (the asm results are only valid behind the wait: nothing touches them before it)
            bf16x4 ra[2][MI], rb[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 32 * ks + 16 * t + 4 * sg + q;
                const int sw = tn_swz(row);
                const unsigned rbase = lds0 + row * 256 + 8 * (pp & 1);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const unsigned addr = rbase + 16 * ((8 * wr + 2 * i + (pp >> 1)) ^ sw);
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ra[t][i]) : "v"(addr));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned addr = rbase + 16384 + 16 * ((8 * wc + 2 * j + (pp >> 1)) ^ sw);
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(rb[t][j]) : "v"(addr));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 wf[4], af[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = bf16x8{ra[0][i][0], ra[0][i][1], ra[0][i][2], ra[0][i][3], ra[1][i][0], ra[1][i][1], ra[1][i][2], ra[1][i][3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = bf16x8{rb[0][j][0], rb[0][j][1], rb[0][j][2], rb[0][j][3], rb[1][j][0], rb[1][j][1], rb[1][j][2], rb[1][j][3]};
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (BF16) {
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
                    } else {
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wf[j]), __builtin_bit_cast(half8, af[i]), acc[j][i], 0, 0, 0);
                    }
                }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");          // MFMA results -> VALU / stores (see fused_ring.hip)
    // lane holds C[m][n .. n + 3]: m = m0 + 64 wr + 16 i + r16, n = n0 + 64 wc + 16 j + 4 kg
    float* cs = p.c + (long)split * p.M * p.ldc;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + 64 * wr + 16 * i + r16;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 64 * wc + 16 * j + 4 * kg;
            if (n + 3 < p.N) *reinterpret_cast<f32x4*>(cs + (long)m * p.ldc + n) = acc[j][i];
            else
                for (int e = 0; e < 4 && n + e < p.N; ++e) cs[(long)m * p.ldc + n + e] = acc[j][i][e];
        }
    }
}

// ---- row-wise LayerNorm over the token stream, 16-byte accesses (E % 8 == 0, E <= 1536) -----------------------------
//   out = (1 - alpha) * src + alpha * (LN(x) * gamma + beta);   x: fp16 | bf16 | f32 [M, ldx];  src: bf16 [M, E] or NULL;
//   out: fp16 | bf16 [M, E].   SigLIP head layernorm (encoder.py:284) and the adaptor blend (projector.py:533-534).
struct LnStreamParams {
    const void* x; int x_dt; long ldx;
    const uint16_t* gamma; const uint16_t* beta;
    const uint16_t* src;
    const void* alpha; int alpha_dt;
    float eps;
    void* out; int out_f16;
    int M, E;
};

__device__ __forceinline__ void ld8(const void* base, int dt, long off, float (&v)[8]) {
    if (dt == HICOM_DT_F32) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else if (dt == HICOM_DT_BF16) {
        const u32x4 g = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16lo_to_f32(g[i]); v[2 * i + 1] = bf16hi_to_f32(g[i]); }
    } else {
        const half8 h = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(base) + off);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
    }
}

__global__ __launch_bounds__(256) void ln_stream_kernel(LnStreamParams p) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int nch = p.E >> 3;
    float v[3][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) ld8(p.x, p.x_dt, m * p.ldx + 8 * ch, v[c]);
        else
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[c][i];
    }
    const float mean = wave_sum_fast(sum) / (float)p.E;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (lane + 64 * c < nch)
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; var = fmaf(d, d, var); }
    const float rstd = 1.0f / sqrtf(wave_sum_fast(var) / (float)p.E + p.eps);
    float alpha = 1.0f;
    if (p.alpha) alpha = p.alpha_dt == HICOM_DT_F32 ? *reinterpret_cast<const float*>(p.alpha) : bf16_to_f32(*reinterpret_cast<const uint16_t*>(p.alpha));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ch = lane + 64 * c;
        if (ch >= nch) continue;
        float g[8], b[8], y[8];
        ld8(p.gamma, HICOM_DT_BF16, 8 * ch, g);
        ld8(p.beta, HICOM_DT_BF16, 8 * ch, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (v[c][i] - mean) * rstd * g[i] + b[i];
        if (p.src) {
            float sv[8];
            ld8(p.src, HICOM_DT_BF16, m * (long)p.E + 8 * ch, sv);
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = (1.0f - alpha) * sv[i] + alpha * y[i];
        }
        if (p.out_f16) {
            half8 h;
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = (_Float16)fminf(fmaxf(y[i], -65504.f), 65504.f);
            *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(p.out) + m * (long)p.E + 8 * ch) = h;
        } else {
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = f32_to_bf16(y[2 * i]) | ((uint32_t)f32_to_bf16(y[2 * i + 1]) << 16);
            *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + m * (long)p.E + 8 * ch) = o;
        }
    }
}

// out[m, :] = x[m, :] / ||x[m, :]||_2 over the token stream (bf16 in, bf16 out, 16-byte accesses): frames_embed of the clip-scale
// local stage when the k adaptor follows (reference projector.py:527-529 in front of :533).
__global__ __launch_bounds__(256) void l2norm_stream_kernel(const uint16_t* x, uint16_t* out, long M, int E) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nch = E >> 3;
    float v[3][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) ld8(x, HICOM_DT_BF16, m * (long)E + 8 * ch, v[c]);
        else
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ss = fmaf(v[c][i], v[c][i], ss);
    }
    const float inv = 1.0f / sqrtf(wave_sum_fast(ss));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ch = lane + 64 * c;
        if (ch >= nch) continue;
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f32_to_bf16(v[c][2 * i] * inv) | ((uint32_t)f32_to_bf16(v[c][2 * i + 1] * inv) << 16);
        *reinterpret_cast<u32x4*>(out + m * (long)E + 8 * ch) = o;
    }
}

}  // namespace hicom

using namespace hicom;

static int dense16_launch(const void* a2, const void* w2, const void* b2, void* out2_f16, void* pre2_f16,
                          const void* a, int64_t lda, const void* w, int64_t ldw, int32_t operand_dt,
                                      const void* b, int32_t b_dt, int32_t M, int32_t N, int32_t K, int32_t act,
                                      void* out_f16, int64_t ldo, int32_t n_store, void* pre_f16, int64_t ldpre,
                                      void* y, int32_t y_dt, int64_t ldy, const void* res, int64_t ldr,
                                      float* ssq, const float* row_tab, int64_t row_tab_ld, int32_t tab_H, int32_t tab_W,
                                      int32_t tab_t0, int32_t tab_y0, int32_t tab_x0,
                                      const void* dot_vec, int32_t dot_vec_dt, float* row_dot, void* stream) {
    HICOM_REQUIRE(a && w && (out_f16 || y || ssq || row_dot), HICOM_EINVAL, "dense16_gemm: NULL pointer / no output");
    if (row_tab) HICOM_REQUIRE(tab_H > 0 && tab_W > 0 && row_tab_ld >= N && row_tab_ld % 4 == 0 && (uintptr_t)row_tab % 16 == 0 &&
                                   M % (tab_H * tab_W) == 0, HICOM_EINVAL, "dense16_gemm: positional table layout");
    HICOM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, HICOM_EINVAL,
                  "dense16_gemm: bad shape M=%d N=%d K=%d (K %% 64, leading dimensions %% 8)", M, N, K);
    HICOM_REQUIRE(operand_dt == HICOM_DT_BF16 || operand_dt == HICOM_DT_F16, HICOM_EINVAL, "dense16_gemm: operands are bf16 or fp16");
    HICOM_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)w % 16 == 0), HICOM_EINVAL, "dense16_gemm: alignment");
    HICOM_REQUIRE((long)M * lda < (1L << 32) && (long)N * ldw < (1L << 32), HICOM_EUNSUP, "dense16_gemm: operand larger than 2^32 elements");
    HICOM_REQUIRE(N % 4 == 0, HICOM_EUNSUP, "dense16_gemm: N %% 4");
    if (out_f16) HICOM_REQUIRE(ldo % 4 == 0 && n_store % 4 == 0 && n_store >= N && n_store <= ldo && (uintptr_t)out_f16 % 8 == 0 &&
                                   n_store <= ((N + 127) / 128) * 128 + 128, HICOM_EINVAL, "dense16_gemm: fp16 output layout");
    if (y) HICOM_REQUIRE(ldy % 4 == 0 && ldy >= N && (uintptr_t)y % 16 == 0 && (!res || (ldr % 4 == 0 && (uintptr_t)res % 8 == 0)), HICOM_EINVAL,
                         "dense16_gemm: packed output layout");
    DenseParams p;
    p.a = (const uint16_t*)a; p.w = (const uint16_t*)w; p.lda = lda; p.ldw = ldw; p.b = b; p.b_f32 = b_dt == HICOM_DT_F32;
    p.M = M; p.N = N; p.K = K; p.act = act; p.o16 = (_Float16*)out_f16; p.ldo = ldo; p.n_store = out_f16 ? n_store : 0;
    p.y = y; p.y_f32 = y_dt == HICOM_DT_F32; p.ldy = ldy; p.res = (const uint16_t*)res; p.ldr = ldr; p.ssq = ssq;
    p.tab = row_tab; p.tab_ld = row_tab_ld; p.H = tab_H; p.W = tab_W; p.t0 = tab_t0; p.y0 = tab_y0; p.x0 = tab_x0;
    // the row-contiguous epilogue moves 8 columns per lane: 16-byte accesses on every output / bias / residual row
    const bool rows = N % 8 == 0 && (!b || (uintptr_t)b % 16 == 0) &&
                      (!out_f16 || (ldo % 8 == 0 && n_store % 8 == 0 && (uintptr_t)out_f16 % 16 == 0)) &&
                      (!y || y_dt == HICOM_DT_F32 || ldy % 8 == 0) && (!res || (ldr % 8 == 0 && (uintptr_t)res % 16 == 0));
    if (row_dot)
        HICOM_REQUIRE(rows && dot_vec && (uintptr_t)dot_vec % 16 == 0 && (dot_vec_dt == HICOM_DT_BF16 || dot_vec_dt == HICOM_DT_F32) &&
                          (!res || ((uintptr_t)res % 16 == 0 && ldr % 8 == 0)), HICOM_EINVAL,
                      "dense16_gemm: the row-dot output needs the row-contiguous epilogue (N %% 8, 16-byte aligned rows) and a bf16 | f32 vector");
    p.dotv = dot_vec; p.dotv_f32 = dot_vec_dt == HICOM_DT_F32; p.rdot = row_dot;
    if (pre_f16)
        HICOM_REQUIRE(rows && ldpre >= N && ldpre % 8 == 0 && (uintptr_t)pre_f16 % 16 == 0, HICOM_EINVAL,
                      "dense16_gemm: the pre-activation output needs the row-contiguous epilogue (N %% 8) and 16-byte aligned rows");
    p.pre16 = (_Float16*)pre_f16; p.ldpre = ldpre;
    hipStream_t st = (hipStream_t)stream;
    const bool bf = operand_dt == HICOM_DT_BF16;
    p.tiles_m = (M + 127) / 128; p.tiles_n = (N + 127) / 128;
    constexpr int smem = (128 + 128) * 128;
    p.a2 = (const uint16_t*)a2; p.w2 = (const uint16_t*)w2; p.b2 = b2; p.o16_2 = (_Float16*)out2_f16; p.pre16_2 = (_Float16*)pre2_f16;
    if (a2) {
        HICOM_REQUIRE(w2 && out_f16 && out2_f16 && !y && !ssq && !row_dot && !row_tab && (!b == !b2) && (!pre_f16 == !pre2_f16), HICOM_EINVAL,
                      "dense16_gemm_pair: two problems of one shape with fp16 outputs (bias / pre-activation output for both or neither)");
        HICOM_REQUIRE(((uintptr_t)a2 % 16 == 0) && ((uintptr_t)w2 % 16 == 0) && ((uintptr_t)out2_f16 % 16 == 0) && (!b2 || (uintptr_t)b2 % 16 == 0) &&
                          (!pre2_f16 || (uintptr_t)pre2_f16 % 16 == 0) && (uintptr_t)out_f16 % 16 == 0 && (!b || (uintptr_t)b % 16 == 0), HICOM_EINVAL,
                      "dense16_gemm_pair: 16-byte alignment");
    }
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * (a2 ? 2 : 1)));
    if (pre_f16 && bf) hipLaunchKernelGGL((dense16_gemm_kernel<true, true, true>), grid, dim3(256), smem, st, p);
    else if (pre_f16) hipLaunchKernelGGL((dense16_gemm_kernel<false, true, true>), grid, dim3(256), smem, st, p);
    else if (bf && rows) hipLaunchKernelGGL((dense16_gemm_kernel<true, true>), grid, dim3(256), smem, st, p);
    else if (bf) hipLaunchKernelGGL((dense16_gemm_kernel<true, false>), grid, dim3(256), smem, st, p);
    else if (rows) hipLaunchKernelGGL((dense16_gemm_kernel<false, true>), grid, dim3(256), smem, st, p);
    else hipLaunchKernelGGL((dense16_gemm_kernel<false, false>), grid, dim3(256), smem, st, p);
    return hicom_host::check_launch("dense16_gemm");
}

extern "C" int hicom_dense16_gemm_fwd(const void* a, int64_t lda, const void* w, int64_t ldw, int32_t operand_dt,
                                      const void* b, int32_t b_dt, int32_t M, int32_t N, int32_t K, int32_t act,
                                      void* out_f16, int64_t ldo, int32_t n_store, void* pre_f16, int64_t ldpre,
                                      void* y, int32_t y_dt, int64_t ldy, const void* res, int64_t ldr,
                                      float* ssq, const float* row_tab, int64_t row_tab_ld, int32_t tab_H, int32_t tab_W,
                                      int32_t tab_t0, int32_t tab_y0, int32_t tab_x0,
                                      const void* dot_vec, int32_t dot_vec_dt, float* row_dot, void* stream) {
    return dense16_launch(nullptr, nullptr, nullptr, nullptr, nullptr, a, lda, w, ldw, operand_dt, b, b_dt, M, N, K, act, out_f16, ldo, n_store, pre_f16, ldpre,
                          y, y_dt, ldy, res, ldr, ssq, row_tab, row_tab_ld, tab_H, tab_W, tab_t0, tab_y0, tab_x0, dot_vec, dot_vec_dt, row_dot, stream);
}

// TWO problems of one shape in ONE launch: out_k = act(a_k . w_k^T + b_k), out_v = act(a_v . w_v^T + b_v) -- the same layer of the k and
// of the v adaptor MLP (reference projector.py:533-534: two independent MLPs over all tokens).  3 285 tiles fill the chip's ~1 000
// resident workgroup slots 3.2 times, i.e. a fifth of the last round is idle; 6 570 tiles share that tail between both problems.
extern "C" int hicom_dense16_gemm_pair_fwd(const void* a_k, const void* w_k, const void* b_k, void* out_k, void* pre_k,
                                           const void* a_v, const void* w_v, const void* b_v, void* out_v, void* pre_v,
                                           int64_t lda, int64_t ldw, int32_t operand_dt, int32_t b_dt, int32_t M, int32_t N, int32_t K, int32_t act,
                                           int64_t ldo, int32_t n_store, int64_t ldpre, void* stream) {
    HICOM_REQUIRE(a_v && w_v && out_v, HICOM_EINVAL, "dense16_gemm_pair: NULL pointer");
    return dense16_launch(a_v, w_v, b_v, out_v, pre_v, a_k, lda, w_k, ldw, operand_dt, b_k, b_dt, M, N, K, act, out_k, ldo, n_store, pre_k, ldpre,
                          nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, 0, 0, nullptr, 0, nullptr, stream);
}

extern "C" int hicom_ln_stream_fwd(const void* x, int32_t x_dt, int64_t ldx, const void* gamma, const void* beta,
                                   const void* src, const void* alpha, int32_t alpha_dt, float eps,
                                   void* out, int32_t out_dt, int32_t M, int32_t E, void* stream) {
    HICOM_REQUIRE(x && gamma && beta && out, HICOM_EINVAL, "ln_stream: NULL pointer");
    HICOM_REQUIRE(M > 0 && E > 0 && E % 8 == 0 && E <= 1536 && ldx >= E && ldx % 8 == 0, HICOM_EINVAL, "ln_stream: bad shape");
    HICOM_REQUIRE(out_dt == HICOM_DT_BF16 || out_dt == HICOM_DT_F16, HICOM_EINVAL, "ln_stream: out is bf16 or fp16");
    HICOM_REQUIRE(src || !alpha, HICOM_EINVAL, "ln_stream: alpha without a blend source");
    LnStreamParams p{x, x_dt, (long)ldx, (const uint16_t*)gamma, (const uint16_t*)beta, (const uint16_t*)src, alpha, alpha_dt, eps,
                     out, out_dt == HICOM_DT_F16, M, E};
    hipLaunchKernelGGL(ln_stream_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return hicom_host::check_launch("ln_stream");
}

extern "C" int hicom_l2norm_stream_fwd(const void* x, void* out, int64_t M, int32_t E, void* stream) {
    HICOM_REQUIRE(x && out && M > 0 && E > 0 && E % 8 == 0 && E <= 1536, HICOM_EINVAL, "l2norm_stream: bad arguments (E %% 8, E <= 1536)");
    HICOM_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0), HICOM_EINVAL, "l2norm_stream: 16-byte alignment");
    hipLaunchKernelGGL(l2norm_stream_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)out,
                       (long)M, E);
    return hicom_host::check_launch("l2norm_stream");
}


extern "C" int hicom_dense16_tn_splits(int32_t M, int32_t N, int64_t Kt) {
    if (M <= 0 || N <= 0 || Kt <= 0) return HICOM_EINVAL;
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128), stages = (Kt + 63) / 64;
    long s = (768 + tiles - 1) / tiles;                  // ~3 workgroups per CU ...
    if (s > stages) s = stages;
    if (stages >= 8) {                                   // ... in multiples of 8 where the token axis allows: one XCD per slice (see the kernel;
        s = (s + 4) / 8 * 8;                             // 3 slices of a 306-tile product ran at half the rate of 8)
        if (s < 8) s = 8;
        if (s > stages) s = stages / 8 * 8;
    }
    if (s > 64) s = 64;
    return (int)(s < 1 ? 1 : s);
}

extern "C" int hicom_dense16_tn_fwd(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t operand_dt, int64_t Kt,
                                    int32_t M, int32_t N, float* c_parts, int64_t ldc, int32_t splits, void* stream) {
    HICOM_REQUIRE(a && b && c_parts, HICOM_EINVAL, "dense16_tn: NULL pointer");
    HICOM_REQUIRE(operand_dt == HICOM_DT_BF16 || operand_dt == HICOM_DT_F16, HICOM_EINVAL, "dense16_tn: operands are both fp16 or both bf16");
    HICOM_REQUIRE(M >= 8 && N >= 8 && Kt > 0 && Kt < (1L << 31) && splits > 0 && splits <= (Kt + 63) / 64 && lda >= M && ldb >= N && ldc >= N, HICOM_EINVAL,
                  "dense16_tn: bad shape");
    HICOM_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && M % 8 == 0 && N % 8 == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) &&
                      ((uintptr_t)c_parts % 16 == 0), HICOM_EINVAL, "dense16_tn: M, N, leading dimensions multiples of 8 elements, 16-byte aligned bases");
    DenseTnParams p{(const uint16_t*)a, (const uint16_t*)b, (long)lda, (long)ldb, M, N, (int)Kt, c_parts, (long)ldc, (M + 127) / 128, (N + 127) / 128, splits};
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * splits));
    if (operand_dt == HICOM_DT_BF16) HICOM_LAUNCH(dense16_tn_kernel<true>, grid, dim3(256), 32768, (hipStream_t)stream, p);
    else HICOM_LAUNCH(dense16_tn_kernel<false>, grid, dim3(256), 32768, (hipStream_t)stream, p);
    return hicom_host::check_launch("dense16_tn");
}
