// Readout MLP GEMM on ONE fp16 plane, with a co-scheduled GEMV role:  y = act(a . w^T + b).
//
// Replaces nn.Linear / nn.GELU / nn.Linear of build_mlp on the window tokens (reference projector.py:307-312, :559)
// on the hot path.  Round 1 carried the fp32 window contexts / hidden activations between kernels as TWO bf16 planes
// (hi + lo) and issued every MFMA twice.  An fp16 plane keeps 11 significand bits -- the rounding of an activation
// costs <= 2^-12 relative, ~5e-5 absolute on these outputs, inside the 1e-3 parity budget -- at HALF the operand
// bytes and HALF the MFMAs; bf16 weights convert to fp16 exactly (8 significand bits; |w| < 6.1e-5 becomes an fp16
// subnormal with <= 3e-8 absolute error), once per weight version (the caller caches the fp16 copies, the way the
// reference caches its pos_embed buffer).  Activations are clamped to the fp16 range when they are written.
//
// The problem is small (1296 x 896 outputs) and is bound by what ONE CU pulls through its vector-memory path
// (~70 GB/s from L2 by LDS-DMA, MI355X_MICROARCH.md "Indexed rows"), so the tile is chosen to minimise the bytes a CU
// stages: 96 x 64 outputs per 256-thread workgroup = (96 + 64) x 128 B = 20 KB per BK = 64 stage (the 48 x 128 hi/lo
// tile of round 1: 28 KB), 14 x 14 = 196 workgroups, one per CU.  Six-stage LDS-DMA ring (global_load_lds_dwordx4,
// counted vmcnt, raw s_barrier), XOR-swizzled [rows][128 B] images, fragment reads of stage s+1 under the MFMAs of
// stage s, product computed transposed (W fragment as the A operand) so that a lane stores 4 consecutive columns.
//
// Round 5: the tile shape is a template parameter (r16_tile<ring, MODE, TN, TM>): 96 x 64 (8-stage ring, the hot path), 96 x 128
// (5 stages) and 192 x 128 (4 stages) for wide layers whose 64-column grid would be several rounds of one-per-CU workgroups
// (hidden 3584); the host takes the smallest tile among those with the fewest rounds.
//
// Horizontal fusion: the launch may carry `n_aux` extra workgroups that run a ROLE on the CUs the tiles leave idle: a single-row
// linear layer (GEMV) of the global compressor's tail (out_proj / readout of the 32 global rows, projector.py:226,646), the merge
// of the ring kernel's partial global states + v_proj (HICOM_ROLE_MERGE_VPROJ), or the tail's two dependent single-row layers with
// an in-launch hand-off (HICOM_ROLE_GEMV_CHAIN): the dependent small launches of the step disappear under the two GEMMs
// (DESIGN.md §3.1).  hicom_readout_tail_fwd (opt-in) runs both GEMMs and both roles in ONE grid.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "common.hpp"
#include "merge_item.hpp"

namespace hicom {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct AuxGemv {
    // y[n] = act(sum_k w[n,k] x[k] + b[n]) + res[n],   x[k] = sum_{s < x_parts} xs[s * x_stride + k] + xb[k]
    const float* xs;
    int x_parts;
    long x_stride;
    const uint16_t* xb;     // bf16 [K] added to x (the bias of the producing layer), or NULL
    const void* w;          // bf16 or f32 [N, K]
    const void* b;          // bf16 or f32 [N] or NULL
    const uint16_t* res;    // bf16 [N] or NULL
    int N, K, act;
    float* y;               // f32 [N] or NULL
    int w_f32, b_f32;
    // optional packed destination: the result row replicated to rows row0 .. row0 + reps - 1 of dst [*, ldd] (the 32
    // identical global rows of "direct" mode, projector.py:646,707)
    void* dst;
    int dst_f32, reps;
    long ldd, row0;
    const long long* x_fixed;   // alternative x: fixed-point accumulators (value * HICOM_FIXED_SCALE), or NULL
    int x_clear;                // GEMV_CHAIN: role workgroup 0 clears x_fixed once every workgroup has read it
};

struct R16Params {
    const _Float16* a;
    const _Float16* w;
    const void* b;
    int b_f32;
    int M, N, K, act;
    _Float16* o16;          // fp16 plane [M][N] (hidden activations for the next GEMM), or NULL
    void* y;                // packed rows of the final tensor (dtype y_f32 ? f32 : bf16), or NULL
    int y_f32;
    long ldy, row0;
    int nl_group;
    int vec, bvec;
    int line16;             // exactly one 16-bit output whose tile rows are whole 16-byte aligned 128-byte lines: the row-line epilogue
    int tn, tm;             // tile shape the host chose (96 x 64 | 96 x 128 | 192 x 128): selects the kernel instantiation
    int n_gemm;             // workgroups of the tile grid; blocks >= n_gemm run the role
    int role;               // HICOM_ROLE_*: what the workgroups behind the tile grid do
    AuxGemv aux;            // GEMV; first layer of GEMV_CHAIN
    AuxGemv aux2;           // GEMV_CHAIN: second layer (x = the first layer's result, handed over inside the launch)
    unsigned* chain_state;  // GEMV_CHAIN: [0..1] 64-bit arrival counter, [2] failed hand-offs, [3] role workgroups of the first launch; granules from byte 256
    MergeVprojFixParams mv; // MERGE_VPROJ
    int cg_cpw1, cg_cpw2, cg_r1, cg_r2;   // GEMV_CHAIN: chain_geom() of this launch, computed by the host
};

// Dev-only timeline (tools/tail_trace.py builds a second library with -DHICOM_TRACE): s_memrealtime (100 MHz, chip-wide) stamps of
// every workgroup of the LAST launch: [0] entry, [1..6] role phases / tile phases, [7] exit.  Compiled out of the product.
#ifdef HICOM_TRACE
__device__ unsigned long long g_r16_trace[512 * 16];
#define R16_TR(k) do { if (threadIdx.x == 0) g_r16_trace[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define R16_TR_XCC() do { if (threadIdx.x == 0) { unsigned x_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x_)); g_r16_trace[blockIdx.x * 16 + 14] = 1000u + (x_ & 15); } } while (0)
#else
#define R16_TR(k) do {} while (0)
#define R16_TR_XCC() do {} while (0)
#endif

constexpr int kRM = 96, kRN = 64;
constexpr int kRImgA = kRM * 128;                   // 12 KB: 96 rows x 64 fp16
constexpr int kRStage = kRImgA + kRN * 128;         // 20 KB
constexpr int kMvRoleItems = 3;                      // merge items a role workgroup keeps in flight together

__device__ __forceinline__ _Float16 to_f16_sat(float v) {
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    return (_Float16)v;
}

// ---- aux role: one GEMV over `n_aux` workgroups.  A wave owns up to CB = 6 columns (1152 columns over 56 x 4 waves = one
// batch) and requests ALL of their weight rows first; the partial vectors of x are requested right behind them, so the two
// cold-memory latencies of the role overlap instead of adding up.
template <bool WF32, bool XFIX>
__device__ __forceinline__ void aux_gemv_role(const AuxGemv& g, int aux_idx, int n_aux, char* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* xl = reinterpret_cast<float*>(lds);       // [K]   x
    float* xp = xl + 1536;                            // [x_parts][K] staged partial vectors
    constexpr int CH = 3;                             // K <= 1536: up to 3 chunks of 8 elements per lane
    constexpr int CB = 6;
    const int nwaves = n_aux * 4, w_id = aux_idx * 4 + wave;
    const int per = (g.N + nwaves - 1) / nwaves;      // columns per wave (contiguous)
    u32x4 wv[CB][CH];                    // bf16 weights: 8 per chunk
    u32x4 wv2[WF32 ? CB : 1][CH];        // f32 weights: the chunk's second four
    constexpr bool wf32 = WF32;
    auto load_w = [&](int n0) {
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int n = n0 + j < g.N ? n0 + j : g.N - 1;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = (lane + 64 * c) * 8;
                if constexpr (wf32) {
                    const float* wr = reinterpret_cast<const float*>(g.w) + (long)n * g.K + k;
                    wv[j][c] = (k < g.K) ? *reinterpret_cast<const u32x4*>(wr) : u32x4{0, 0, 0, 0};
                    wv2[j][c] = (k < g.K) ? *reinterpret_cast<const u32x4*>(wr + 4) : u32x4{0, 0, 0, 0};
                } else {
                    wv[j][c] = (k < g.K) ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(g.w) + (long)n * g.K + k) : u32x4{0, 0, 0, 0};
                }
            }
        }
    };
    const int n_first = w_id * per;
    load_w(n_first < g.N ? n_first : 0);
    // x = sum of the partial vectors (+ bias).  Phase 1: every float4 of every part is requested at once (a serial
    // loop over the parts is x_parts dependent L2 round trips: 20 us for 18 parts) and parked in LDS; phase 2 sums
    // each column over the parts in part order -- deterministic, no atomics.
    if constexpr (XFIX) {
        // x from the fixed-point accumulators of hicom_merge_vproj_fixed_fwd: ONE 8-byte value per element (K <= 1536 = 6 per thread)
        long long xf[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) xf[u] = (tid + 256 * u < g.K) ? g.x_fixed[tid + 256 * u] : 0ll;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int k = tid + 256 * u;
            if (k < g.K) xl[k] = (float)xf[u] * (1.0f / HICOM_FIXED_SCALE) + (g.xb ? bf16_to_f32(g.xb[k]) : 0.f);
        }
    } else {
        const int k4n = g.K >> 2, items = g.x_parts * k4n;
        constexpr int XR = 28;                            // x_parts * K / 4 <= 28 * 256 (host-checked)
        float4 t[XR];
        {
            int sidx = tid / k4n, k4 = tid - sidx * k4n;                        // (part, float4 column) of item tid; then += 256 items
            const int dq = 256 / k4n, dr = 256 - dq * k4n;
#pragma unroll
            for (int u = 0; u < XR; ++u) {
                const bool in = tid + 256 * u < items;                           // (clamped: the loads are unconditional)
                t[u] = *reinterpret_cast<const float4*>(g.xs + (long)(in ? sidx : 0) * g.x_stride + 4 * (in ? k4 : 0));
                sidx += dq;
                k4 += dr;
                if (k4 >= k4n) { k4 -= k4n; ++sidx; }
            }
        }
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            const int it = tid + 256 * u;
            if (it < items) *reinterpret_cast<float4*>(xp + 4 * it) = t[u];     // [part][K] order: it = part * k4n + k4
        }
        __syncthreads();
        for (int k = tid; k < g.K; k += 256) {
            float v = g.xb ? bf16_to_f32(g.xb[k]) : 0.f;
            int sidx = 0;
            for (; sidx + 6 <= g.x_parts; sidx += 6) {                          // six LDS reads in flight; summed in part order
                float r[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) r[u] = xp[(sidx + u) * g.K + k];
#pragma unroll
                for (int u = 0; u < 6; ++u) v += r[u];
            }
            for (; sidx < g.x_parts; ++sidx) v += xp[sidx * g.K + k];
            xl[k] = v;
        }
    }
    __syncthreads();
    float xr[CH][8];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int k = (lane + 64 * c) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[c][i] = (k < g.K) ? xl[k + i] : 0.f;
    }
    const int n_end = min(g.N, n_first + per);
    for (int n0 = n_first; n0 < n_end; n0 += CB) {
        if (n0 != n_first) load_w(n0);
        float vout[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (wf32) {
                        acc = fmaf(__uint_as_float(wv[j][c][i]), xr[c][i], acc);
                        acc = fmaf(__uint_as_float(wv2[j][c][i]), xr[c][4 + i], acc);
                    } else {
                        acc = fmaf(bf16lo_to_f32(wv[j][c][i]), xr[c][2 * i], acc);
                        acc = fmaf(bf16hi_to_f32(wv[j][c][i]), xr[c][2 * i + 1], acc);
                    }
                }
            acc = wave_sum_fast(acc);                                        // (every lane holds the sum)
            const int n = n0 + j < g.N ? n0 + j : g.N - 1;
            float v = acc;
            if (g.b) v += g.b_f32 ? reinterpret_cast<const float*>(g.b)[n] : bf16_to_f32(reinterpret_cast<const uint16_t*>(g.b)[n]);
            if (g.act == HICOM_ACT_GELU) v = gelu_erf(v);
            if (g.res) v += bf16_to_f32(g.res[n]);
            vout[j] = v;
            if (lane == 0 && n0 + j < n_end && g.y) g.y[n] = v;
        }
        if (g.dst) {
            // lane r writes this batch's columns of replica row r
            for (int r = lane; r < g.reps; r += 64) {
                const long o = (g.row0 + r) * g.ldd + n0;
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    if (n0 + j < n_end) {
                        if (g.dst_f32) reinterpret_cast<float*>(g.dst)[o + j] = vout[j];
                        else reinterpret_cast<uint16_t*>(g.dst)[o + j] = f32_to_bf16(vout[j]);
                    }
            }
        }
    }
}


// ---- GEMV_CHAIN role (round 5): TWO dependent single-row layers in one launch,
//   h[n]  = act1(sum_k w1[n,k] x[k] + b1[n]),   x from the fixed-point accumulators of the merge (+ xb)
//   y2[m] = act2(sum_n w2[m,n] h[n] + b2[m])    -> the replicated rows of the output
// i.e. GELU(C o + r0) and the last global readout layer (reference projector.py:226, :646, :307-312 after folding): the whole
// global tail behind the merge rides under readout GEMM 2, and the merge itself under GEMM 1 -- its launch is gone from the step.
// The N1 <= 1536 intermediate values travel as 8-byte {epoch, f32} granules (cdna_hip_programming.md Guideline 16, form R2; the
// scheme of query_prep.hip): every role workgroup first produces its columns of h, then sweeps ALL granules.  The second layer's
// weight rows are requested before anything else, so their cold-memory latency runs under the first layer.
// Residency: the role workgroups have the highest block indices; the tile workgroups never wait, so every role workgroup is
// dispatched at the latest when the tiles have finished -- a spinning role workgroup cannot starve its producers as long as the
// role fits the chip (n_role <= CUs; host-checked).  Every spin is bounded; a failed hand-off poisons the result with NaN.
typedef __attribute__((address_space(1))) unsigned long long r16_gu64;
typedef __attribute__((address_space(1))) unsigned int r16_gu32;

// Layout of the role's LDS (the launch's 160 KB, unused by a role workgroup otherwise): weight rows of both layers land by LDS-DMA --
// a role CU that loads into REGISTERS keeps ~64 KB in flight (26 GB/s beside the streaming tiles: tools/tail_trace.py saw the last
// of 111 KB requested 4.6 us after entry), the DMA path keeps everything it is given in flight (the tiles' own 140 KB per CU).
constexpr int kChainStateHead = 256;              // state words (arrival counter: an atomic per role workgroup) on lines of their own, granules behind
constexpr int kChainLds = 160 * 1024, kChainVec = 1536 * 4, kChainBias = 2 * 256 * 4, kChainOut = 256 * 4;
constexpr int kChainW2Max = 40 * 1024;             // second-layer rows resident per batch (all of a workgroup's at the release shape: 16 x 1792 B)
// Round 6: hidden layers up to 4096 wide (the 7B model's 3584, BASELINE configs[3]): the vector area holds 4096 floats, a lane sweeps up to
// 16 granules, a second-layer row is up to 8 KB.  The first batch of second-layer rows is prefetched as before; once the first layer is
// through, its weight region is free and the later batches take both regions (19 rows of 7 KB instead of 5 per round trip).
constexpr int kChainBigN = 4096, kChainVecBig = kChainBigN * 4;

struct ChainGeom { int cpw1, cpw2, r1, r2; long row1, row2; };
__host__ __device__ inline long chain_pad1k(long b) { return (b + 1023) & ~1023L; }
__host__ __device__ inline int chain_vec_bytes(int N1) { return N1 > 1536 ? kChainVecBig : kChainVec; }
__host__ __device__ inline ChainGeom chain_geom(int N1, int K1, bool w1f32, int N2, int an) {
    ChainGeom g;
    g.cpw1 = (N1 + an - 1) / an;
    g.cpw2 = (N2 + an - 1) / an;
    g.row1 = (long)K1 * (w1f32 ? 4 : 2);
    g.row2 = (long)N1 * 2;
    g.r2 = (int)(kChainW2Max / g.row2) < g.cpw2 ? (int)(kChainW2Max / g.row2) : g.cpw2;
    // (a batch lands in whole 1-KB pieces: each region is rounded up to the piece size, so a tail piece never reaches the next region)
    const long left = kChainLds - chain_vec_bytes(N1) - kChainBias - kChainOut - chain_pad1k((long)g.r2 * g.row2) - 1024;
    g.r1 = (int)(left / g.row1) < g.cpw1 ? (int)(left / g.row1) : g.cpw1;
    return g;
}

// `mid`: work in front of the role's first use of x (the fused tail launch: the merge items that PRODUCE x and the wait for every role
// workgroup's; it may use the first kChainVec + kChainBias + kChainOut bytes of `lds`).  With a gated mid the order is: mid's requests,
// mid's work and arrival at the gate, THEN the weight rows (waves 1..3: wave 0 polls the gate, and a wave's requests return in order),
// the gate, x with agent-scope loads.  A round trip beside ~200 streaming tile workgroups is 2-4 us, so the role is priced in dependent
// round trips: the weight rows first made the merge -- the head of the chain -- wait for ~100 KB per CU that nothing needs before the
// gate has passed (tools/tail3_trace.py: merge done at 12.7 us instead of 10.5).  Tried and dropped: accumulators that carry their own
// completion count, polled by every role workgroup in place of the gate -- 62 k polling loads per round on the 72 lines the atomic
// adds are working on: x complete at 24 us instead of 14.6.
struct ChainNoMid {
    __device__ __forceinline__ void pre() {}                                   // (the mid work's requests)
    __device__ __forceinline__ void reduce() {}                                // (the mid work itself, up to this workgroup's arrival at the gate)
    __device__ __forceinline__ bool wait() { return false; }                   // (the gate; returns whether x must be treated as lost)
};
// What the chain role reads BEHIND its first requests.  LATE (the kernel whose preloaded leading arguments are the role's, PRE = 1): filled from the
// kernel-argument block through a pointer the compiler cannot see through before that point -- otherwise hipcc fetches these fields at the
// role's entry and waits for them in front of the first request (an SGPR it spills there is enough).
struct ChainLate {
    const uint16_t* xb;
    const void *b1, *b2;
    const uint16_t *res1, *res2;
    float *y1, *y2;
    void* dst;
    long ldd, row0;
    int b1_f32, b2_f32, act1, act2, dst_f32, reps, x_clear, K2;
};
typedef const __attribute__((address_space(4))) R16Params* r16_kernarg_ptr;
template <bool W1F32, class Mid, bool BIG = false, bool LATE = false>
__device__ __forceinline__ void gemv_chain_role(const R16Params& p, int ai, int an, char* lds, Mid& mid, r16_kernarg_ptr late = nullptr) {
    constexpr bool kGated = !std::is_same<Mid, ChainNoMid>::value;
    constexpr int XV = BIG ? kChainBigN : 1536;                       // floats of the vector area (x, later h)
    static_assert(!(BIG && kGated), "the gated (fused tail) form takes the release widths only");
    const AuxGemv& g1 = p.aux;
    const AuxGemv& g2 = p.aux2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    R16_TR(6);   // role entered (kernel arguments read)
    ChainGeom cg;                                                     // (host-computed: four integer divisions off the role's critical path)
    cg.cpw1 = p.cg_cpw1; cg.cpw2 = p.cg_cpw2; cg.r1 = p.cg_r1; cg.r2 = p.cg_r2;
    cg.row1 = (long)g1.K * (W1F32 ? 4 : 2); cg.row2 = (long)g1.N * 2;
    float* xl = reinterpret_cast<float*>(lds);                       // [XV] x, later h
    float* bl1 = xl + XV;                                            // [256] first-layer biases of this workgroup's columns
    float* bl2 = bl1 + 256;                                          // [256] second-layer biases
    float* yl = bl2 + 256;                                           // [256] second-layer results of this workgroup
    char* w2l = lds + XV * 4 + kChainBias + kChainOut;               // [r2][row2]
    char* w1l = w2l + chain_pad1k((long)cg.r2 * cg.row2);            // [r1][row1]
    r16_gu64* cnt = (r16_gu64*)p.chain_state;
    r16_gu64* gran = (r16_gu64*)((char*)p.chain_state + kChainStateHead);
    const int n_lo = ai * cg.cpw1, n_hi = min(g1.N, n_lo + cg.cpw1);  // first-layer columns of this workgroup
    const int m_lo = ai * cg.cpw2, m_hi = min(g2.N, m_lo + cg.cpw2);  // second-layer columns
    // (the counter is requested first and CONSUMED behind the first barrier: a returned atomic takes 2-4 us beside the tiles)
    unsigned long long c0 = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    R16_TR(8);
    // Every request of the prologue goes out BEFORE anything is computed from a returned value: straight-line, unconditional loads
    // (clamped indices instead of predicates, raw bias bits instead of a per-type branch), pinned by a scheduling barrier.  hipcc had
    // placed the int64 -> float conversion of x -- and its wait -- in front of the bias and weight requests, and the bf16 bias branch
    // carried a vmcnt(0) of its own (tools/tail_trace.py: last request 4.5 us after entry).
    long long xf[6];                                                  // x from the fixed-point accumulators (K1 <= 1536 = 6 per thread)
    if constexpr (!kGated) {
#pragma unroll
        for (int u = 0; u < 6; ++u) xf[u] = g1.x_fixed[min(tid + 256 * u, g1.K - 1)];
    }
    R16_TR(9);
    auto bias_raw = [&](const void* b, int b_f32, const void* w, int N, int n) -> unsigned {
        n = n < N ? n : N - 1;
        const char* bp = reinterpret_cast<const char*>(b ? b : w) + (long)n * (b_f32 ? 4 : 2);
        return b_f32 ? *reinterpret_cast<const unsigned*>(bp) : (unsigned)*reinterpret_cast<const uint16_t*>(bp);
    };
    // DMA of rows [r_lo, r_hi) of a row-major matrix (a CONTIGUOUS byte range) into `dst`, 1-KB pieces dealt over the four waves;
    // a piece's tail beyond the range re-reads the matrix's last 16 bytes (lands in unread LDS)
    auto dma_rows = [&](const void* w, long rowbytes, long total_rows, int r_lo, int r_hi, char* dst) -> int {
        if (r_hi <= r_lo) return 0;
        const long lo = (long)r_lo * rowbytes, bytes = (long)(r_hi - r_lo) * rowbytes, last = total_rows * rowbytes - 16;
        const int npieces = (int)((bytes + 1023) >> 10);
        int mine = 0;
        for (int pi = kGated ? wave - 1 : wave; pi < npieces; pi += kGated ? 3 : 4, ++mine) {
            if (kGated && wave == 0) break;
            long off = lo + (long)pi * 1024 + lane * 16;
            off = off < last ? off : last;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(w) + off),
                                             (__attribute__((address_space(3))) void*)(dst + (long)pi * 1024), 16, 0, 0);
        }
        return mine;
    };
    R16_TR(10);
    if constexpr (kGated) {
        mid.pre();
        mid.reduce();
    }
    dma_rows(g1.w, cg.row1, g1.N, n_lo, min(n_hi, n_lo + cg.r1), w1l);
    R16_TR(11);
    dma_rows(g2.w, cg.row2, g2.N, m_lo, min(m_hi, m_lo + cg.r2), w2l);
    __builtin_amdgcn_sched_barrier(0);
    // (the biases LAST: their pointers are not among the preloaded kernel arguments -- everything above leaves without waiting for the argument block)
    ChainLate L;
    if constexpr (LATE) {
        asm volatile("" : "+s"(late));
        L.xb = late->aux.xb; L.b1 = late->aux.b; L.b2 = late->aux2.b; L.res1 = late->aux.res; L.res2 = late->aux2.res; L.y1 = late->aux.y; L.y2 = late->aux2.y;
        L.dst = late->aux2.dst; L.ldd = late->aux2.ldd; L.row0 = late->aux2.row0; L.b1_f32 = late->aux.b_f32; L.b2_f32 = late->aux2.b_f32;
        L.act1 = late->aux.act; L.act2 = late->aux2.act; L.dst_f32 = late->aux2.dst_f32; L.reps = late->aux2.reps; L.x_clear = late->aux.x_clear; L.K2 = late->aux2.K;
    } else {
        L.xb = g1.xb; L.b1 = g1.b; L.b2 = g2.b; L.res1 = g1.res; L.res2 = g2.res; L.y1 = g1.y; L.y2 = g2.y;
        L.dst = g2.dst; L.ldd = g2.ldd; L.row0 = g2.row0; L.b1_f32 = g1.b_f32; L.b2_f32 = g2.b_f32;
        L.act1 = g1.act; L.act2 = g2.act; L.dst_f32 = g2.dst_f32; L.reps = g2.reps; L.x_clear = g1.x_clear; L.K2 = g2.K;
    }
    const unsigned b1raw = bias_raw(L.b1, L.b1_f32, g1.w, g1.N, n_lo + tid), b2raw = bias_raw(L.b2, L.b2_f32, g2.w, g2.N, m_lo + tid);
    __builtin_amdgcn_sched_barrier(0);
    R16_TR(1);   // every load requested
    bool xlost = false;
    if constexpr (kGated) {
        xlost = mid.wait();
#pragma unroll
        for (int u = 0; u < 6; ++u)
            xf[u] = (long long)__hip_atomic_load((r16_gu64*)(g1.x_fixed + min(tid + 256 * u, g1.K - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int k = tid + 256 * u;
        if (k < g1.K) xl[k] = xlost ? __uint_as_float(0x7FC00000u) : (float)xf[u] * (1.0f / HICOM_FIXED_SCALE) + (L.xb ? bf16_to_f32(L.xb[k]) : 0.f);
    }
    bl1[tid] = L.b1 ? __uint_as_float(L.b1_f32 ? b1raw : b1raw << 16) : 0.f;
    bl2[tid] = L.b2 ? __uint_as_float(L.b2_f32 ? b2raw : b2raw << 16) : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's DMA pieces (both layers') have landed
    __syncthreads();
    R16_TR(2);   // x and the weights landed
    asm volatile("" : "+v"(c0));                                      // nothing computed from the counter is scheduled above this point
    const unsigned epoch = (unsigned)(c0 / (unsigned)an) + 1u;
    // dot products of up to FOUR LDS-resident weight rows (this wave's: rows r, r + 4, r + 8, r + 12 of the batch) with the vector in
    // xl, their chains interleaved: lane owns elements 4 lane + 256 c (f32 rows) or 8 lane + 512 c (bf16 rows)
    auto dots_f32 = [&](const char* row0, long stride, int nrows, int K, float (&out)[4]) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int k = 4 * lane + 256 * c;
            if (k < K) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xl + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nrows) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(row0 + j * stride + (long)k * 4);
                        acc[j] = fmaf(wv[0], xv[0], acc[j]); acc[j] = fmaf(wv[1], xv[1], acc[j]);
                        acc[j] = fmaf(wv[2], xv[2], acc[j]); acc[j] = fmaf(wv[3], xv[3], acc[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = wave_sum_fast(acc[j]);
    };
    auto dots_bf16 = [&](const char* row0, long stride, int nrows, int K, float (&out)[4]) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < (BIG ? 8 : 3); ++c) {
            const int k = 8 * lane + 512 * c;
            if (k < K) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xl + k), x1 = *reinterpret_cast<const f32x4*>(xl + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nrows) {
                        const u32x4 wv = *reinterpret_cast<const u32x4*>(row0 + j * stride + (long)k * 2);
                        acc[j] = fmaf(bf16lo_to_f32(wv[0]), x0[0], acc[j]); acc[j] = fmaf(bf16hi_to_f32(wv[0]), x0[1], acc[j]);
                        acc[j] = fmaf(bf16lo_to_f32(wv[1]), x0[2], acc[j]); acc[j] = fmaf(bf16hi_to_f32(wv[1]), x0[3], acc[j]);
                        acc[j] = fmaf(bf16lo_to_f32(wv[2]), x1[0], acc[j]); acc[j] = fmaf(bf16hi_to_f32(wv[2]), x1[1], acc[j]);
                        acc[j] = fmaf(bf16lo_to_f32(wv[3]), x1[2], acc[j]); acc[j] = fmaf(bf16hi_to_f32(wv[3]), x1[3], acc[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = wave_sum_fast(acc[j]);
    };
    // ---- first layer: rows of the resident batch dealt over the four waves; further batches (shapes whose rows do not fit at once) ----
    for (int b0 = n_lo; b0 < n_hi; b0 += cg.r1) {
        const int b1 = min(n_hi, b0 + cg.r1);
        if (b0 != n_lo) {
            __syncthreads();                                          // the previous batch has been consumed
            dma_rows(g1.w, cg.row1, g1.N, b0, b1, w1l);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        for (int n = b0 + wave; n < b1; n += 16) {                     // rows n, n + 4, n + 8, n + 12 of this wave, together
            const int nrows = min(4, (b1 - n + 3) >> 2);
            float d[4];
            if (W1F32) dots_f32(w1l + (long)(n - b0) * cg.row1, 4 * cg.row1, nrows, g1.K, d);
            else dots_bf16(w1l + (long)(n - b0) * cg.row1, 4 * cg.row1, nrows, g1.K, d);
            float v = lane == 0 ? d[0] : lane == 1 ? d[1] : lane == 2 ? d[2] : d[3];          // lane j < nrows finishes row n + 4 j
            const int nn = n + 4 * lane;
            if (lane < nrows) {
                v += bl1[nn - n_lo];
                if (L.act1 == HICOM_ACT_GELU) v = gelu_erf(v);
                if (L.res1) v += bf16_to_f32(L.res1[nn]);
                __hip_atomic_store(gran + nn, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                if (L.y1) L.y1[nn] = v;
            }
        }
    }
    // every wave is done with x in xl: an LDS-only meeting.  __syncthreads() also drains vmcnt, i.e. waits for the acknowledgement of the
    // granule stores just issued -- a round trip (2-4 us beside the tiles) that nothing here needs: the sweep's polls go out at once and
    // return behind those acknowledgements anyway (a wave's requests complete in order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    R16_TR(3);   // first layer done, granules stored
    // ---- sweep: wave w collects granules [w * q, (w + 1) * q) of h until every tag carries this launch's epoch ----
    {
        constexpr int GPL = BIG ? 16 : 6;             // granules per lane and wave: N1 <= 4 * 64 * GPL = 1536 (4096)
        const int q = (g1.N + 3) >> 2, lo = wave * q, hi = min(g1.N, lo + q);
        unsigned long long gv[GPL];
        unsigned spins = 0;
        const unsigned grid0 = __hip_atomic_load((r16_gu32*)p.chain_state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool failed = grid0 != 0u && grid0 != (unsigned)an;       // (the epoch arithmetic needs the same role size on every launch)
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < GPL; ++u) {
                const int n = lo + lane + 64 * u;
                gv[u] = (n < hi) ? __hip_atomic_load(gran + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)epoch << 32);
                ok &= (unsigned)(gv[u] >> 32) == epoch;
            }
            if (__all(ok)) break;
            if (failed || ++spins > (1u << 22)) {
                failed = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (failed && lane == 0) atomicAdd(p.chain_state + 2, 1u);
#pragma unroll
        for (int u = 0; u < GPL; ++u) {
            const int n = lo + lane + 64 * u;
            if (n < hi) xl[n] = failed ? __uint_as_float(0x7FC00000u) : __uint_as_float((unsigned)gv[u]);
        }
    }
    __syncthreads();
    R16_TR(4);   // hand-off complete
    // (every granule carries this launch's epoch: every role workgroup has published, i.e. has read x -- the accumulators can be
    // cleared for the next step's merge; a failed hand-off has poisoned this launch anyway and the next one starts behind it)
    if (L.x_clear && ai == 0)
        for (int k = tid; k < g1.K; k += 256) const_cast<long long*>(g1.x_fixed)[k] = 0ll;
    // ---- second layer (bf16 rows, K2 = N1) ----
    // (BIG: behind the first, prefetched batch the first layer's weight region is free too -- later batches take [w2l, end of the LDS))
    const int r2b = BIG ? max(cg.r2, (int)((kChainLds - (XV * 4 + kChainBias + kChainOut) - 1024) / cg.row2)) : cg.r2;
    for (int b0 = m_lo, b1 = min(m_hi, m_lo + cg.r2); b0 < m_hi; b0 = b1, b1 = min(m_hi, b0 + r2b)) {
        if (b0 != m_lo) {
            __syncthreads();
            dma_rows(g2.w, cg.row2, g2.N, b0, b1, w2l);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        for (int m = b0 + wave; m < b1; m += 16) {
            const int nrows = min(4, (b1 - m + 3) >> 2);
            float d[4];
            dots_bf16(w2l + (long)(m - b0) * cg.row2, 4 * cg.row2, nrows, L.K2, d);
            float v = lane == 0 ? d[0] : lane == 1 ? d[1] : lane == 2 ? d[2] : d[3];
            const int mm = m + 4 * lane;
            if (lane < nrows) {
                v += bl2[mm - m_lo];
                if (L.act2 == HICOM_ACT_GELU) v = gelu_erf(v);
                if (L.res2) v += bf16_to_f32(L.res2[mm]);
                yl[mm - m_lo] = v;
                if (L.y2) L.y2[mm] = v;
            }
        }
    }
    __syncthreads();
    if (L.dst) {
        // the workgroup's columns of every replica row: consecutive threads write consecutive columns
        const int ncol = m_hi - m_lo;
        for (int idx = tid; idx < L.reps * ncol; idx += 256) {
            const int rr = idx / ncol, c = idx - rr * ncol;
            const long o = (L.row0 + rr) * L.ldd + m_lo + c;
            if (L.dst_f32) reinterpret_cast<float*>(L.dst)[o] = yl[c];
            else reinterpret_cast<uint16_t*>(L.dst)[o] = f32_to_bf16(yl[c]);
        }
    }
    // the arrival: after every wave of the workgroup has read the counter (all did before the first barrier)
    R16_TR(5);   // second layer done, rows stored
    if (tid == 0 && ai == 0 && __hip_atomic_load((r16_gu32*)p.chain_state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        __hip_atomic_store((r16_gu32*)p.chain_state + 3, (unsigned)an, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool W1F32>
__device__ __forceinline__ void gemv_chain_role(const R16Params& p, int ai, int an, char* lds) {
    ChainNoMid none;
    gemv_chain_role<W1F32, ChainNoMid>(p, ai, an, lds, none);
}

// ---- MERGE_VPROJ role: the (head, 64-channel slab) items of hicom_merge_vproj_fixed_fwd dealt over the role workgroups ----
__device__ __forceinline__ void merge_vproj_role(const R16Params& p, int ai, int an, char* lds) {
    const int nslab = p.mv.E / 64, rows = p.mv.E / p.mv.hd, items = nslab * rows;
    // up to kMvRoleItems items per workgroup with ALL their loads requested before the first is reduced (host: items <= kMvRoleItems * an)
    MvItemRegs<64, true> regs[kMvRoleItems];
#pragma unroll
    for (int u = 0; u < kMvRoleItems; ++u) {
        const int it = ai + u * an;
        if (it < items) merge_vproj_fixed_load<64, true>(p.mv, it % nslab, it / nslab, regs[u]);
    }
#pragma unroll
    for (int u = 0; u < kMvRoleItems; ++u) {
        const int it = ai + u * an;
        if (it < items) merge_vproj_fixed_compute<64, true>(p.mv, it % nslab, it / nslab, regs[u], lds);
    }
    for (int it = ai + kMvRoleItems * an; it < items; it += an)        // (never at the shapes the host admits; kept for safety)
        merge_vproj_fixed_item<64, true>(p.mv, it % nslab, it / nslab, lds);
}

template <int N, int I = 0, class F>
__device__ __forceinline__ void r16_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        r16_static_for<N, I + 1>(f);
    }
}
template <int N>
__device__ __forceinline__ void r16_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


enum { R16_PLAIN = 0, R16_PUBLISH = 1, R16_CONSUME = 2 };
typedef __attribute__((address_space(1))) unsigned long long r16_sync64;
// Counters of the fused tail launch (hicom_readout_tail_fwd), each on a 128-byte line of its own; all cumulative over launches:
// a launch's epoch = arrivals / gridDim.x + 1 (every workgroup adds one arrival when it is done).
struct TileSync {
    unsigned long long* arrivals;     // [0] arrivals, [1] failed waits
    unsigned long long* gate;         // the merge role's fan-in (one add per role workgroup)
    unsigned long long* rowblk;       // [nby][16]: tiles of row block `by` published (one add per tile, `per_epoch` per launch)
    unsigned long long* flags;        // [tiles]: (epoch << 1 | abandoned) once GEMM 2's tile has an owner or has been given up by it
    int per_epoch;
};

// the workgroup -> tile map of the tile grid: ordinal of the tile in the blocked order below, or -1 (the grid is rounded up to 8 x slots)
template <int TN = kRN, int TM = kRM>
__device__ __forceinline__ int r16_tile_of_block(const R16Params& p, int vblock) {
    const int nbx = (p.N + TN - 1) / TN, nby = (p.M + TM - 1) / TM;
    // XCD-balanced order (speed only): workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md "Workgroup dispatch"); every
    // XCD gets a contiguous run of ~tiles/8 tiles in row-major tile order, i.e. ~2 row tiles whose A rows stay in ITS L2,
    // and -- the point -- the same number of busy CUs, so that the aux workgroups (dealt round-robin too) find a free CU
    // on their XCD at once instead of queueing behind a 10-us tile (measured: +6 us per launch with whole row tiles per XCD)
    const int xcd = vblock & 7, slot = vblock >> 3;
    const int tiles = nbx * nby;
    const int t_lo = (tiles * xcd) >> 3, t_hi = (tiles * (xcd + 1)) >> 3;
    return slot >= t_hi - t_lo ? -1 : t_lo + slot;
}

// One TM x TN output tile of y = act(a . w^T + b) by one 256-thread workgroup (`tile` = r16_tile_of_block<TN, TM>()).
// TN = 64: the hot path's tile (20 KB per BK = 64 stage, 8-stage ring).  TN = 128 (round 5): for wide layers -- hidden 3584, the 7B
// model's width -- where the 64-column grid is several rounds of workgroups: half the tiles, A staged half as often (28 KB per
// stage, 5-stage ring); a wave owns 48 x 64 outputs.  TM = 192 (with TN = 128: 40-KB stages, 4-stage ring, a wave owns 96 x 64 outputs): when
// even the 96 x 128 grid is more than one round (1296 rows at hidden 3584: 392 tiles -> 196).
// MODE: R16_PLAIN; R16_PUBLISH = the fp16 plane leaves as write-through stores and the tile's row block counts it (the fused tail
// launch: the plane is the next GEMM's A operand, read inside the same launch); R16_CONSUME = the A rows are such a plane: the W
// stages of the prologue go out at entry, the A stages behind the row block's counter (`epoch` = this launch's).  A consumer that
// `may_abandon` waits a BOUNDED time: if the row block is not complete by then it marks the tile abandoned and returns false (it
// must not hold its CU: a publisher may still be waiting for one); otherwise it marks the tile owned.  Returns true when the tile is done.
template <int kRRing, int MODE, int TN = kRN, int TM = kRM>
__device__ __forceinline__ bool r16_tile(const R16Params& p, int tile, char* lds, const TileSync sy, unsigned long long epoch = 0, bool may_abandon = false) {
    static_assert((TN == 64 || TN == 128) && (TM == 96 || (TM == 192 && TN == 128)), "tile shape");
    constexpr int NJ = TN / 32;                              // 16-column blocks per wave (a wave owns TN / 2 columns)
    constexpr int MI = TM / 32;                              // 16-row blocks per wave (a wave owns TM / 2 rows)
    constexpr int AW = TM / 32;                              // A pieces per wave and stage (TM / 8 one-KiB pieces over four waves)
    constexpr int PW = AW + TN / 32;                         // DMA pieces (1 KiB = 8 rows x 128 B) per wave and stage: TM / 8 A + TN / 8 W over four waves
    constexpr int IMG_A = TM * 128;                          // bytes of a stage's A image
    constexpr int STAGE = IMG_A + TN * 128;                  // bytes of a ring stage
    static_assert(kRRing * STAGE <= 160 * 1024, "ring exceeds the LDS");
    constexpr int kTrOff = MODE == R16_CONSUME ? 8 : 0;      // (dev timeline: a consumer's stamps go to slots of their own)
    (void)kTrOff;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, kg = lane >> 4;
    const int nbx = (p.N + TN - 1) / TN, nby = (p.M + TM - 1) / TM;
    // ... in a BLOCKED order: the left half of the tile columns row by row, then the right half -- an XCD's run of ~tiles/8 tiles is a
    // ~(3.5 rows x 7 columns) patch, so both its A row blocks and its W column blocks are shared by several of its CUs and cross
    // its L2 once (row-major runs shared the A rows 14 ways and the W columns hardly at all: every CU pulled its W panel from
    // beyond L2, profiles/r04_d: 24.5 MB of fabric traffic for 7.3 MB of operands)
    int by, bx;
    {
        const int hx = nbx >= 4 ? (nbx + 1) >> 1 : nbx, nleft = nby * hx;
        if (tile < nleft) {
            by = tile / hx;
            bx = tile - by * hx;
        } else {
            const int t2 = tile - nleft, wx = nbx - hx;
            by = t2 / wx;
            bx = hx + t2 - by * wx;
        }
    }
    const int m0 = by * TM, n0 = bx * TN;
    const int ns = p.K >> 6;

    // DMA assignment: a stage is TM / 8 + TN / 8 one-KiB pieces (8 rows x 128 B): A first, then W; wave w issues pieces w, w+4, ...
    const int prow = lane >> 3, cpos = lane & 7;
    const _Float16* src[PW];
    int dst_off[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int pi = wave + 4 * i;
        if (pi < TM / 8) {
            const int row = 8 * pi + prow;
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            src[i] = p.a + (long)m * p.K + 8 * (cpos ^ ((row >> 1) & 7));
        } else {
            const int row = 8 * (pi - TM / 8) + prow;
            int n = n0 + row;
            n = n < p.N ? n : p.N - 1;
            src[i] = p.w + (long)n * p.K + 8 * (cpos ^ ((row >> 1) & 7));
        }
        dst_off[i] = pi * 1024;
    }
    // (this wave's first AW pieces are A rows, the others W rows: wave + 4 i < TM / 8 <=> i < AW.)  A published plane (R16_CONSUME) is read with sc1
    // loads: they bypass this CU's L1, the one cache a write-through store of another CU does not reach (MI355X_MICROARCH.md
    // "inter-workgroup visibility": every store of the bytes sc1 and drained, every load of them sc1 -- no acquire needed)
    constexpr int kAuxA = MODE == R16_CONSUME ? 16 : 0;
    auto issue_a = [&](int s, int ring_slot) {
        char* base = lds + ring_slot * STAGE;
#pragma unroll
        for (int i = 0; i < AW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + 64 * s),
                                             (__attribute__((address_space(3))) void*)(base + dst_off[i]), 16, 0, kAuxA);
    };
    auto issue_w = [&](int s, int ring_slot) {
        char* base = lds + ring_slot * STAGE;
#pragma unroll
        for (int i = AW; i < PW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + 64 * s),
                                             (__attribute__((address_space(3))) void*)(base + dst_off[i]), 16, 0, 0);
    };
    auto issue = [&](int s, int ring_slot) {
        issue_a(s, ring_slot);
        issue_w(s, ring_slot);
    };
    auto wait_stages = [&](int k) {      // at most k of this wave's stages still in flight
        if (k >= 6) r16_wait_vm<6 * PW>();
        else if (k == 5) r16_wait_vm<5 * PW>();
        else if (k == 4) r16_wait_vm<4 * PW>();
        else if (k == 3) r16_wait_vm<3 * PW>();
        else if (k == 2) r16_wait_vm<2 * PW>();
        else if (k == 1) r16_wait_vm<PW>();
        else r16_wait_vm<0>();
    };

    // fragments of one BK = 64 stage: two K = 32 steps x (NJ W blocks of this wave's TN / 2 columns, MI A blocks of its TM / 2 rows)
    struct Frags {
        half8 w[2][NJ], a[2][MI];
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(lds);
    const int swz = (r16 >> 1) & 7;
    const int f0off = r16 * 128 + 16 * (kg ^ swz), f1off = r16 * 128 + 16 * ((4 + kg) ^ swz);
    const int a_base = (TM / 2) * wr * 128, w_base = IMG_A + (TN / 2) * wc * 128;
#define HICOM_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto read = [&](int ring_slot, Frags& f) {
        const unsigned st = lds0 + ring_slot * STAGE;
        const unsigned a0 = st + a_base + f0off, a1 = st + a_base + f1off, w0 = st + w_base + f0off, w1 = st + w_base + f1off;
        HICOM_LDS_RD(f.w[0][0], w0, 0);
        HICOM_LDS_RD(f.w[0][1], w0, 2048);
        if constexpr (NJ == 4) {
            HICOM_LDS_RD(f.w[0][NJ - 2], w0, 4096);
            HICOM_LDS_RD(f.w[0][NJ - 1], w0, 6144);
        }
        HICOM_LDS_RD(f.a[0][0], a0, 0);
        HICOM_LDS_RD(f.a[0][1], a0, 2048);
        HICOM_LDS_RD(f.a[0][2], a0, 4096);
        if constexpr (MI == 6) {
            HICOM_LDS_RD(f.a[0][MI - 3], a0, 6144);
            HICOM_LDS_RD(f.a[0][MI - 2], a0, 8192);
            HICOM_LDS_RD(f.a[0][MI - 1], a0, 10240);
        }
        HICOM_LDS_RD(f.w[1][0], w1, 0);
        HICOM_LDS_RD(f.w[1][1], w1, 2048);
        if constexpr (NJ == 4) {
            HICOM_LDS_RD(f.w[1][NJ - 2], w1, 4096);
            HICOM_LDS_RD(f.w[1][NJ - 1], w1, 6144);
        }
        HICOM_LDS_RD(f.a[1][0], a1, 0);
        HICOM_LDS_RD(f.a[1][1], a1, 2048);
        HICOM_LDS_RD(f.a[1][2], a1, 4096);
        if constexpr (MI == 6) {
            HICOM_LDS_RD(f.a[1][MI - 3], a1, 6144);
            HICOM_LDS_RD(f.a[1][MI - 2], a1, 8192);
            HICOM_LDS_RD(f.a[1][MI - 1], a1, 10240);
        }
    };
#undef HICOM_LDS_RD
    auto land = [&](Frags& f) {
        if constexpr (MI == 6)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.w[0][NJ - 2]), "+v"(f.w[0][NJ - 1]), "+v"(f.w[1][0]), "+v"(f.w[1][1]),
                           "+v"(f.w[1][NJ - 2]), "+v"(f.w[1][NJ - 1]), "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[0][2]), "+v"(f.a[0][MI - 3]),
                           "+v"(f.a[0][MI - 2]), "+v"(f.a[0][MI - 1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[1][2]), "+v"(f.a[1][MI - 3]),
                           "+v"(f.a[1][MI - 2]), "+v"(f.a[1][MI - 1])::"memory");
        else if constexpr (NJ == 4)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.w[0][NJ - 2]), "+v"(f.w[0][NJ - 1]), "+v"(f.w[1][0]), "+v"(f.w[1][1]),
                           "+v"(f.w[1][NJ - 2]), "+v"(f.w[1][NJ - 1]), "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[0][2]), "+v"(f.a[1][0]),
                           "+v"(f.a[1][1]), "+v"(f.a[1][2])::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.w[1][0]), "+v"(f.w[1][1]), "+v"(f.a[0][0]), "+v"(f.a[0][1]),
                           "+v"(f.a[0][2]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[1][2])::"memory");
    };
    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const Frags& f) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[ks][j], f.a[ks][i], acc[j][i], 0, 0, 0);
    };
    // The stage's MFMAs with the NEXT ring stage's DMA pieces issued BETWEEN them.  A wave is alone on its SIMD: nothing but its own instruction
    // stream overlaps anything, and a DMA instruction (64 lanes x 16 B) holds the CU's address unit for ~16 clocks -- with four waves issuing their
    // pieces together right behind the barrier, a wave sat ~320-460 clocks in front of its MFMAs (tools/r16_hot_trace.py: 740 clocks per 64-deep
    // step on cache-hot operands, the same as in the step: the K loop was bound by its own instruction order, not by memory).  Interleaved, the
    // address unit works under the matrix pipe: GAP MFMAs, one piece, GAP MFMAs, one piece, ...  (sched_barrier pins the order: hipcc's
    // scheduler would regroup them).
    auto issue_piece = [&](auto I, int s_issue, char* base) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + 64 * s_issue),
                                         (__attribute__((address_space(3))) void*)(base + dst_off[i]), 16, 0, i < AW ? kAuxA : 0);
    };
    // (The fragment reads of stage s + 1 stay in ONE group in front of the MFMAs.  tools/r16_hot_trace.py with the loop taken apart, cache-hot
    // operands, clocks per 64-deep step of the 96 x 64 tile: barrier + 10 ds_read_b128 + their landing 440, + the 12 MFMAs 175, + the 5 DMA pieces
    // 110 = 725.  One read per MFMA gap as well bought another 7-13 % of the K loop -- but with reads between the MFMAs hipcc rotates the
    // accumulator blocks through the AGPRs from step to step (20 v_accvgpr moves per step), and the way around that, every MFMA as inline asm, takes
    // the accumulators out of the hazard recogniser's sight: a compiler copy of an accumulator behind an MFMA in flight showed up at the loop exits
    // and one parity case failed.  Not kept: profiles/r06_q_r16_loop.txt.)
    auto compute_issue = [&](const Frags& f, Frags& nxt, int slot_rd, int s_issue, int ring_slot, bool do_issue) {
        constexpr int NM = 2 * NJ * MI, GAP = NM / PW > 0 ? NM / PW : 1;
        char* base = lds + ring_slot * STAGE;
        read(slot_rd, nxt);                       // the reads of stage s+1 first: they fly under the MFMAs of stage s
        __builtin_amdgcn_sched_barrier(0);
        r16_static_for<NM>([&](auto M) {
            constexpr int m = decltype(M)::value;
            constexpr int ks = m / (NJ * MI), j = (m / MI) % NJ, i = m % MI;
#ifndef HICOM_R16_DEV_NO_MFMA
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[ks][j], f.a[ks][i], acc[j][i], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m % GAP == GAP - 1 && m / GAP < PW) {
#ifndef HICOM_R16_DEV_NO_DMA
                if (do_issue) issue_piece(std::integral_constant<int, m / GAP>{}, s_issue, base);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // (pieces the MFMA count did not reach: none for the three tile shapes -- NM / GAP >= PW)
        r16_static_for<PW>([&](auto I) {
            if constexpr (decltype(I)::value >= NM / GAP)
                if (do_issue) issue_piece(I, s_issue, base);
        });
    };

    // prologue: stages 0 .. kRRing-2 in flight, stage 0 landed, its fragments on the way
    const int npro = ns < kRRing - 1 ? ns : kRRing - 1;
    bool poisoned = false;
    if constexpr (MODE == R16_CONSUME) {
        // (host: ns >= kRRing + 6, so the prologue is full and the first six steps are steady ones)
        r16_sync64* cnt = (r16_sync64*)(sy.rowblk + 16 * by);
        const unsigned long long target = epoch * (unsigned long long)sy.per_epoch;
        unsigned long long seen = 0;
        if (tid == 0) seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (requested in FRONT of the W pieces: returns first)
        for (int s = 0; s < kRRing - 1; ++s) issue_w(s, s);
        unsigned* flag = reinterpret_cast<unsigned*>(lds + (kRRing - 1) * STAGE);   // (the ring's last slot: first written by step 0)
        if (tid == 0) {
            // an owner that may abandon waits ~100 us at most (every publisher of a healthy launch is done within ~2); a sweeper
            // (every publisher has finished by the time it runs) waits like every other hand-off of the step
            const unsigned limit = may_abandon ? 256u : (1u << 22);
            unsigned spins = 0, failed = 0;
            while (seen < target) {
                if (++spins > limit) {
                    failed = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (may_abandon)
                __hip_atomic_store((r16_sync64*)(sy.flags + tile), (epoch << 1) | failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (failed)
                __hip_atomic_fetch_add((r16_sync64*)sy.arrivals + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = failed;
        }
        __syncthreads();
        poisoned = *flag != 0u;
        R16_TR(6 + kTrOff);   // consumer: row block published
        if (poisoned && may_abandon) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the W pieces land in LDS this workgroup is about to give back)
            __syncthreads();
            return false;
        }
        for (int s = 0; s < kRRing - 1; ++s) issue_a(s, s);
    } else {
        for (int s = 0; s < npro; ++s) issue(s, s);
    }
    // the bias of this wave's columns, fetched now (one vector load per column block)
    uint2 braw[NJ];
    float4 brawf[NJ];
    bool bpre[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        braw[j] = make_uint2(0, 0);
        brawf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int nb = n0 + (TN / 2) * wc + 16 * j + 4 * kg;
        bpre[j] = p.b && p.bvec && nb + 3 < p.N;
        if (bpre[j]) {
            if (p.b_f32) brawf[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.b) + nb);
            else braw[j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.b) + nb);
        }
    }
    if constexpr (MODE == R16_CONSUME) r16_wait_vm<(kRRing - 2) * 3>();          // A pieces of stage 0 landed (every W piece is older)
    else if (npro == kRRing - 1) r16_wait_vm<(kRRing - 2) * PW>();
    else wait_stages(npro - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    R16_TR(1 + kTrOff);   // tile: first stage landed
    Frags f0, f1;
    read(0, f0);
    land(f0);
    int slot_next = 1;                 // ring slot of stage s+1
    int slot_issue = kRRing - 1;       // ring slot of stage s+kRRing-1 (= the slot of stage s-1)
    auto step = [&](auto steady, int s, const Frags& cur, Frags& nxt) {
#ifdef HICOM_R16_SERIAL_ISSUE
        // (round 2-6 form, dev A/B: every piece of the next ring stage right behind the barrier, then the reads, then the MFMAs)
        if constexpr (decltype(steady)::value) {
            r16_wait_vm<(kRRing - 3) * PW>();         // stage s+1 landed (this wave's pieces); s+2 .. s+kRRing-2 may fly
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(s + kRRing - 1, slot_issue);
        } else {
            const int ahead = (ns - 1 < s + kRRing - 2 ? ns - 1 : s + kRRing - 2) - (s + 1);
            wait_stages(ahead);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + kRRing - 1 < ns) issue(s + kRRing - 1, slot_issue);
        }
        read(slot_next, nxt);
        __builtin_amdgcn_sched_barrier(0);      // the reads of stage s+1 fly under the MFMAs of stage s
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        land(nxt);
#else
        bool do_issue = true;
        if constexpr (decltype(steady)::value) {
            r16_wait_vm<(kRRing - 3) * PW>();         // stage s+1 landed (this wave's pieces); s+2 .. s+kRRing-2 may fly
        } else {
            const int ahead = (ns - 1 < s + kRRing - 2 ? ns - 1 : s + kRRing - 2) - (s + 1);
            wait_stages(ahead);
            do_issue = s + kRRing - 1 < ns;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        compute_issue(cur, nxt, slot_next, s + kRRing - 1, slot_issue, do_issue);      // MFMAs of stage s with the reads of s+1 and the pieces of s+kRRing-1 in their gaps
        __builtin_amdgcn_sched_barrier(0);
        land(nxt);
#endif
        slot_next = slot_next + 1 == kRRing ? 0 : slot_next + 1;
        slot_issue = slot_issue + 1 == kRRing ? 0 : slot_issue + 1;
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    int s = 0;
    if constexpr (MODE == R16_CONSUME) {
        // the first kRRing - 3 steps wait on a prologue whose A pieces were issued LAST: behind stage s + 1's A pieces are the A pieces
        // of prologue stages s + 2 .. kRRing - 2 (3 each) and the whole stages issued by steps 0 .. s - 1 (PW each)
        auto early = [&](auto allow, int s_, const Frags& cur, Frags& nxt) {
            r16_wait_vm<decltype(allow)::value>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(s_ + kRRing - 1, slot_issue);
            read(slot_next, nxt);
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            land(nxt);
            slot_next = slot_next + 1 == kRRing ? 0 : slot_next + 1;
            slot_issue = slot_issue + 1 == kRRing ? 0 : slot_issue + 1;
        };
        static_assert(kRRing == 8 && TN == 64 && TM == 96, "the early-step waits are written out for an 8-stage ring of 96 x 64 tiles");
        early(std::integral_constant<int, 3 * 5 + PW * 0>{}, 0, f0, f1);
        early(std::integral_constant<int, 3 * 4 + PW * 1>{}, 1, f1, f0);
        early(std::integral_constant<int, 3 * 3 + PW * 2>{}, 2, f0, f1);
        early(std::integral_constant<int, 3 * 2 + PW * 3>{}, 3, f1, f0);
        early(std::integral_constant<int, 3 * 1 + PW * 4>{}, 4, f0, f1);
        step(Yes{}, 5, f1, f0);
        s = 6;
    }
    for (; s + kRRing < ns; s += 2) {
        step(Yes{}, s, f0, f1);
        step(Yes{}, s + 1, f1, f0);
    }
    for (; s + 2 < ns; s += 2) {
        step(No{}, s, f0, f1);
        step(No{}, s + 1, f1, f0);
    }
    if (ns - s == 2) {
        step(No{}, s, f0, f1);
        compute(f1);
    } else {
        compute(f0);
    }
    // MFMA results -> VALU reads (CDNA4 ISA §4.1 "XDL write VGPR -> VALU read": do not rely on hipcc's padding, see fused_ring.hip)
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    R16_TR(2 + kTrOff);   // tile: main loop done
    // Row-line epilogue (one 16-bit output, whole 64-column tiles): the results go through the (idle) ring as a [96][64] 16-bit image
    // and leave as 16-byte stores, eight consecutive lanes writing one 128-byte line of an output row.  The straight form -- every lane
    // storing its four columns of six accumulator blocks, 8 bytes at a time, 16 partial lines per instruction -- is store-ISSUE bound
    // (cdna_hip_programming.md T21; tools/tail_trace.py: 1.8 us for 12 KB per workgroup, the same on a second pass with warm caches).
    if (p.line16 && n0 + TN <= p.N) {
        constexpr int TP = TN + 8;                                 // row pitch in 16-bit elements: 16-byte aligned rows, shifted banks
        uint16_t* tl = reinterpret_cast<uint16_t*>(lds);
        __syncthreads();                                           // every wave has read its last fragments out of the ring
        const bool to_f16 = p.o16 != nullptr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (bpre[j]) {
                if (p.b_f32) {
                    bias[0] = brawf[j].x; bias[1] = brawf[j].y; bias[2] = brawf[j].z; bias[3] = brawf[j].w;
                } else {
                    bias[0] = bf16lo_to_f32(braw[j].x); bias[1] = bf16hi_to_f32(braw[j].x);
                    bias[2] = bf16lo_to_f32(braw[j].y); bias[3] = bf16hi_to_f32(braw[j].y);
                }
            } else if (p.b) {
                const int n = n0 + (TN / 2) * wc + 16 * j + 4 * kg;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bias[q] = p.b_f32 ? reinterpret_cast<const float*>(p.b)[n + q] : bf16_to_f32(reinterpret_cast<const uint16_t*>(p.b)[n + q]);
            }
#pragma unroll
            for (int im = 0; im < MI; ++im) {
                uint16_t h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[j][im][q] + bias[q];
                    if (p.act == HICOM_ACT_GELU) v = gelu_erf(v);
                    if (MODE == R16_CONSUME && poisoned) v = __uint_as_float(0x7FC00000u);     // the row block never arrived: fail loudly
                    if (to_f16) {
                        const _Float16 hv = to_f16_sat(v);
                        h[q] = __builtin_bit_cast(uint16_t, hv);
                    } else {
                        h[q] = f32_to_bf16(v);
                    }
                }
                *reinterpret_cast<uint2*>(tl + ((TM / 2) * wr + 16 * im + r16) * TP + (TN / 2) * wc + 16 * j + 4 * kg) =
                    make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            }
        }
        __syncthreads();
        for (int it = tid; it < TM * (TN / 8); it += 256) {
            const int row = it / (TN / 8), c8 = it % (TN / 8), m = m0 + row;
            if (m < p.M) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(tl + row * TP + 8 * c8);
                if (to_f16) {
                    if constexpr (MODE == R16_PUBLISH)        // write-through: the bytes are at the memory side when the store is acknowledged
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p.o16 + (long)m * p.N + n0 + 8 * c8), "v"(v) : "memory");
                    else
                        *reinterpret_cast<u32x4*>(p.o16 + (long)m * p.N + n0 + 8 * c8) = v;
                } else {
                    const long orow = p.row0 + m + (p.nl_group > 0 ? m / p.nl_group : 0);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.y) + orow * p.ldy + n0 + 8 * c8) = v;
                }
            }
        }
        R16_TR(5 + kTrOff);   // tile: stores issued
        if constexpr (MODE == R16_PUBLISH) {
            // every storing wave drains its stores, the workgroup meets, ONE lane counts the tile in
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add((r16_sync64*)(sy.rowblk + 16 * by), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return true;
    }
    if constexpr (MODE != R16_PLAIN) return true;          // (host: the fused tail launch admits row-line shapes only)
    // epilogue.  Transposed product: lane holds columns n .. n+3 (4 * kg + q) of row m = r16 of each block.
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + (TN / 2) * wc + 16 * j + 4 * kg;
        if (n >= p.N) continue;
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (bpre[j]) {
            if (p.b_f32) {
                bias[0] = brawf[j].x; bias[1] = brawf[j].y; bias[2] = brawf[j].z; bias[3] = brawf[j].w;
            } else {
                bias[0] = bf16lo_to_f32(braw[j].x); bias[1] = bf16hi_to_f32(braw[j].x);
                bias[2] = bf16lo_to_f32(braw[j].y); bias[3] = bf16hi_to_f32(braw[j].y);
            }
        } else if (p.b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nn = n + q < p.N ? n + q : p.N - 1;
                bias[q] = p.b_f32 ? reinterpret_cast<const float*>(p.b)[nn] : bf16_to_f32(reinterpret_cast<const uint16_t*>(p.b)[nn]);
            }
        }
        const bool vec = p.vec && n + 3 < p.N;
#pragma unroll
        for (int im = 0; im < MI; ++im) {
            const int m = m0 + (TM / 2) * wr + 16 * im + r16;
            if (m >= p.M) continue;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = acc[j][im][q] + bias[q];
                if (p.act == HICOM_ACT_GELU) v[q] = gelu_erf(v[q]);
            }
            if (p.o16) {
                _Float16* oh = p.o16 + (long)m * p.N + n;
                if (vec) {
                    *reinterpret_cast<half4*>(oh) = half4{to_f16_sat(v[0]), to_f16_sat(v[1]), to_f16_sat(v[2]), to_f16_sat(v[3])};
                } else {
                    for (int q = 0; q < 4 && n + q < p.N; ++q) oh[q] = to_f16_sat(v[q]);
                }
            }
            if (p.y) {
                const long orow = p.row0 + m + (p.nl_group > 0 ? m / p.nl_group : 0);
                if (p.y_f32) {
                    float* yo = reinterpret_cast<float*>(p.y) + orow * p.ldy + n;
                    if (vec) *reinterpret_cast<float4*>(yo) = make_float4(v[0], v[1], v[2], v[3]);
                    else
                        for (int q = 0; q < 4 && n + q < p.N; ++q) yo[q] = v[q];
                } else {
                    uint16_t* yo = reinterpret_cast<uint16_t*>(p.y) + orow * p.ldy + n;
                    if (vec) {
                        const uint32_t u0 = f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                        const uint32_t u1 = f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(yo) = make_uint2(u0, u1);
                    } else {
                        for (int q = 0; q < 4 && n + q < p.N; ++q) yo[q] = f32_to_bf16(v[q]);
                    }
                }
            }
        }
    }
    R16_TR(5 + kTrOff);   // tile: stores issued
    return true;
}

// The leading scalar arguments repeat what a workgroup's FIRST requests depend on: hipcc preloads them into SGPRs at wave launch (-mllvm
// -amdgpu-kernarg-preload-count, build_native.py; 14 dwords at most), so those requests do not wait for a kernel-argument fetch -- a memory
// round trip beside ~250 workgroups that start together.  Everything else is read from the argument block as before.  WHOSE first requests:
//   PRE = 0  the tiles':  q0 = a, q1 = w, i0..i2 = M, N, K, i3 = n_gemm, i4 = role, i5 = role workgroups        (the tiles end the launch)
//   PRE = 1  the GEMV chain role's (the launch it ends: its x, counter and weight requests leave beside the tiles' first DMA instead of behind
//            31 MB of them):  q0 = chain_state, q1 = aux.x_fixed, q2 = aux.w, q3 = aux2.w, i0 = aux.K, i1 = aux.N | aux2.N << 16,
//            i2 = cg_cpw1 | cg_cpw2 << 16, i3 = n_gemm | role workgroups << 16, i4 = role | aux.w_f32 << 8, i5 = cg_r1 | cg_r2 << 16; the tiles fetch their operand bases
template <int kRRing, int TN = kRN, int TM = kRM, int PRE = 0>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void readout16_gemm_kernel(const void* q0, const void* q1, const void* q2, const void* q3,
                                                                                                            int i0, int i1, int i2, int i3, int i4, int i5, R16Params p_in) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [kRRing][stage]; the roles take the whole 160 KB
    R16Params p = p_in;
    if constexpr (PRE == 0) {
        p.a = (const _Float16*)q0; p.w = (const _Float16*)q1; p.M = i0; p.N = i1; p.K = i2; p.n_gemm = i3; p.role = i4;
    } else {
        p.chain_state = (unsigned*)const_cast<void*>(q0); p.aux.x_fixed = (const long long*)q1; p.aux.w = q2; p.aux2.w = q3;
        p.aux.K = i0; p.aux.N = i1 & 0xffff; p.aux2.N = (int)((unsigned)i1 >> 16);
        p.cg_cpw1 = i2 & 0xffff; p.cg_cpw2 = (int)((unsigned)i2 >> 16); p.cg_r1 = i5 & 0xffff; p.cg_r2 = (int)((unsigned)i5 >> 16);
        p.n_gemm = i3 & 0xffff; p.role = i4 & 0xff; p.aux.w_f32 = i4 >> 8;
    }
    // (the role size rides in the preloaded arguments too: gridDim.x is a fetch from the argument block's hidden tail)
    const int n_aux_pre = PRE == 0 ? i5 : (int)((unsigned)i3 >> 16);
    R16_TR(0);
    R16_TR_XCC();
    if ((int)blockIdx.x >= p.n_gemm) {
        const int ai = (int)blockIdx.x - p.n_gemm, an = n_aux_pre;
        if (p.role == HICOM_ROLE_MERGE_VPROJ) {
            merge_vproj_role(p, ai, an, lds);
            R16_TR(7);
            return;
        }
        if constexpr (PRE == 1) {
            // (the argument block sits behind the 14 preloaded dwords: four pointers + six ints, and it is 8-byte aligned)
            static_assert(4 * sizeof(void*) + 6 * sizeof(int) == 56 && alignof(R16Params) <= 8, "kernel-argument offset of the parameter block");
            r16_kernarg_ptr late = (r16_kernarg_ptr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + 56);
            ChainNoMid none;
            if (p.aux.N > 1536) {
                if (p.aux.w_f32) gemv_chain_role<true, ChainNoMid, true, true>(p, ai, an, lds, none, late);
                else gemv_chain_role<false, ChainNoMid, true, true>(p, ai, an, lds, none, late);
            } else if (p.aux.w_f32) gemv_chain_role<true, ChainNoMid, false, true>(p, ai, an, lds, none, late);
            else gemv_chain_role<false, ChainNoMid, false, true>(p, ai, an, lds, none, late);
            R16_TR(7);
            return;
        }
        if (p.role == HICOM_ROLE_GEMV_CHAIN) {
            if (p.aux.N > 1536) {                 // (hidden 3584: wide vector area, more granules per lane, 8-KB second-layer rows)
                ChainNoMid none;
                if (p.aux.w_f32) gemv_chain_role<true, ChainNoMid, true>(p, ai, an, lds, none);
                else gemv_chain_role<false, ChainNoMid, true>(p, ai, an, lds, none);
            } else if (p.aux.w_f32) gemv_chain_role<true>(p, ai, an, lds);
            else gemv_chain_role<false>(p, ai, an, lds);
            R16_TR(7);
            return;
        }
        if (p.aux.x_fixed) {
            if (p.aux.w_f32) aux_gemv_role<true, true>(p.aux, ai, an, lds);
            else aux_gemv_role<false, true>(p.aux, ai, an, lds);
        } else {
            if (p.aux.w_f32) aux_gemv_role<true, false>(p.aux, ai, an, lds);
            else aux_gemv_role<false, false>(p.aux, ai, an, lds);
        }
        return;
    }
    const int tile = r16_tile_of_block<TN, TM>(p, (int)blockIdx.x);
    if (tile >= 0) r16_tile<kRRing, R16_PLAIN, TN, TM>(p, tile, lds, TileSync{nullptr, nullptr, nullptr, nullptr, 0});
}

// ---- the fused tail launch: both readout GEMMs and the merge -> two-layer chain role in ONE grid ----
// Workgroups [0, n1) are tile workgroups: each computes its tile of GEMM 1 (hidden = GELU(ctx . W1^T + b1)), PUBLISHES it (write-through
// stores, a per-row-block counter), requests the W2 rows of the SAME tile of GEMM 2 (y = hidden . W2^T + b2), waits for the 14 tiles
// of its row block and computes it: no second cold start, no kernel boundary between the two GEMMs.  Workgroups [n1, n1 + n_role)
// run the merge items, meet at a counter, and run the global tail's two single-row layers.
//
// Progress without any assumption about how many workgroups are resident: a tile workgroup's wait for its row block is BOUNDED -- if
// the block is not complete in time (publishers still queueing for a CU: fewer CUs than workgroups, a co-tenant kernel) it marks its
// GEMM-2 tile abandoned and EXITS, so publishers never starve.  The role workgroups (highest block ids: dispatched last) finish by
// waiting until every GEMM-2 tile is owned or abandoned -- every tile workgroup gets there in bounded time -- and computing the
// abandoned ones (none in a healthy launch: one look at the flags, off the critical path).
struct TailParams {
    R16Params g1;           // GEMM 1 (+ mv: the merge items)
    R16Params g2;           // GEMM 2 (+ aux, aux2, chain_state, cg_*: the chain)
    TileSync sy;
    int n1, n_role, tiles;
};

template <int kRRing>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void readout_tail_kernel(TailParams q) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    R16_TR(0);
    const int b = (int)blockIdx.x, tid = threadIdx.x;
    // this launch's epoch, from the arrival counter (every workgroup adds one when it is done: gridDim.x per launch).  Requested by
    // every thread's first instruction, consumed behind the first phase.
    unsigned long long c0 = __hip_atomic_load((r16_sync64*)q.sy.arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool tile_wg = b < q.n1;
    int t2 = -1, t2_step = q.tiles;                // GEMM-2 tiles of this workgroup: t2, t2 + t2_step, ... (a role workgroup: the abandoned ones among them)
    if (tile_wg) {
        const int tile = r16_tile_of_block(q.g1, b);
        if (tile >= 0) {
            r16_tile<kRRing, R16_PUBLISH>(q.g1, tile, lds, q.sy);
            t2 = tile;
        }
    } else {
        const int ai = b - q.n1, an = q.n_role;
        // the merge items' loads go out FIRST (a CU's requests return in order: behind the chain's ~100 KB of weight rows they came back
        // 6 us later, tools/tail3_trace.py), the chain's weight rows right behind them, then the items are reduced
        struct Hooks {
            const TailParams& q;
            char* lds;
            int ai, an;
            const unsigned long long& epoch_src;        // (a reference: the value is consumed behind the merge items, not here)
            MvItemRegs<64, true> regs[kMvRoleItems];
            __device__ __forceinline__ void pre() {
                const int nslab = q.g1.mv.E / 64, items = nslab * (q.g1.mv.E / q.g1.mv.hd);
#pragma unroll
                for (int u = 0; u < kMvRoleItems; ++u) {
                    const int it = ai + u * an;
                    if (it < items) merge_vproj_fixed_load<64, true>(q.g1.mv, it % nslab, it / nslab, regs[u]);
                }
            }
            __device__ __forceinline__ void reduce() {
                const int nslab = q.g1.mv.E / 64, items = nslab * (q.g1.mv.E / q.g1.mv.hd);
#pragma unroll
                for (int u = 0; u < kMvRoleItems; ++u) {
                    const int it = ai + u * an;
                    if (it < items) merge_vproj_fixed_compute<64, true>(q.g1.mv, it % nslab, it / nslab, regs[u], lds);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's atomic adds are at the memory side
                __syncthreads();
                R16_TR(12);   // role: merge items done
                if (threadIdx.x == 0) __hip_atomic_fetch_add((r16_sync64*)q.sy.gate, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __device__ __forceinline__ bool wait() {
                unsigned* lost = reinterpret_cast<unsigned*>(lds + kChainVec + kChainBias);      // (the chain's result area: unused until its second layer)
                if (threadIdx.x == 0) {
                    r16_sync64* gate = (r16_sync64*)q.sy.gate;
                    const unsigned long long target = (epoch_src / gridDim.x + 1ull) * (unsigned long long)an;
                    unsigned spins = 0, failed = 0;
                    while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        if (++spins > (1u << 22)) {
                            // (counted; the chain runs on NaNs so that its own hand-off still completes and the result says so)
                            __hip_atomic_fetch_add((r16_sync64*)q.sy.arrivals + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            failed = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    *lost = failed;
                }
                __syncthreads();
                R16_TR(13);   // role: every merge item of the launch done
                return *lost != 0u;
            }
        };
        Hooks mid{q, lds, ai, an, c0, {}};
        if (q.g2.aux.w_f32) gemv_chain_role<true>(q.g2, ai, an, lds, mid);
        else gemv_chain_role<false>(q.g2, ai, an, lds, mid);
        // ---- sweep: every GEMM-2 tile owned or abandoned?  (thread t looks at tiles t, t + 256, ...) ----
        const unsigned long long epoch = c0 / gridDim.x + 1ull;
        unsigned* cnt_l = reinterpret_cast<unsigned*>(lds);         // [0] abandoned tiles, [1] timed out
        __syncthreads();                                             // (the chain is done with its LDS)
        if (tid < 2) cnt_l[tid] = 0u;
        __syncthreads();
        unsigned spins = 0;
        for (int t = tid; t < q.tiles; t += 256) {
            unsigned long long f;
            while (((f = __hip_atomic_load((r16_sync64*)(q.sy.flags + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != epoch) {
                if (++spins > (1u << 22)) {
                    atomicAdd(&cnt_l[1], 1u);
                    f = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (f & 1ull) atomicAdd(&cnt_l[0], 1u);
        }
        __syncthreads();
        if (cnt_l[1] != 0u && tid == 0) __hip_atomic_fetch_add((r16_sync64*)q.sy.arrivals + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cnt_l[0] != 0u) {                                        // (rare: abandoned tile t belongs to role workgroup t % an)
            t2 = ai;
            t2_step = an;
        }
        __syncthreads();
    }
    // ---- GEMM 2: a tile workgroup's own tile (bounded wait, may abandon); a role workgroup's share of the abandoned tiles ----
    {
        asm volatile("" : "+v"(c0));
        const unsigned long long epoch = c0 / gridDim.x + 1ull;
        for (; t2 >= 0 && t2 < q.tiles; t2 += t2_step) {
            if (!tile_wg) {
                const unsigned long long f = __hip_atomic_load((r16_sync64*)(q.sy.flags + t2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!((f >> 1) == epoch && (f & 1ull))) continue;
                __syncthreads();
            }
            r16_tile<kRRing, R16_CONSUME>(q.g2, t2, lds, q.sy, epoch, tile_wg);
            if (!tile_wg) __syncthreads();
        }
    }
    R16_TR(7);
    if (tid == 0) __hip_atomic_fetch_add((r16_sync64*)q.sy.arrivals, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bf16 -> fp16 (weights, once per weight version) and f32 -> fp16 (saturating) conversions
__global__ __launch_bounds__(256) void to_f16_kernel(const void* src, int src_f32, _Float16* dst, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i + q < n) {
            const float v = src_f32 ? reinterpret_cast<const float*>(src)[i + q] : bf16_to_f32(reinterpret_cast<const uint16_t*>(src)[i + q]);
            dst[i + q] = to_f16_sat(v);
        }
}

__global__ __launch_bounds__(256) void to_f16_padded_kernel(const void* src, int src_f32, long rows, long cols, _Float16* dst, long ld) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ld) return;
    const long r = i / ld, c = i - r * ld;
    float v = 0.f;
    if (c < cols) v = src_f32 ? reinterpret_cast<const float*>(src)[r * cols + c] : bf16_to_f32(reinterpret_cast<const uint16_t*>(src)[r * cols + c]);
    dst[i] = to_f16_sat(v);
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_to_f16_padded_fwd(const void* src, int32_t src_dt, int64_t rows, int64_t cols, void* dst, int64_t ld_dst, void* stream) {
    HICOM_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_dst >= cols, HICOM_EINVAL, "to_f16_padded: bad arguments");
    hipLaunchKernelGGL(to_f16_padded_kernel, dim3((unsigned)((rows * ld_dst + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       src_dt == HICOM_DT_F32, (long)rows, (long)cols, (_Float16*)dst, (long)ld_dst);
    return hicom_host::check_launch("to_f16_padded");
}

extern "C" int hicom_to_f16_fwd(const void* src, int32_t src_dt, void* dst, int64_t n, void* stream) {
    HICOM_REQUIRE(src && dst && n > 0, HICOM_EINVAL, "to_f16: bad arguments");
    hipLaunchKernelGGL(to_f16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, src,
                       src_dt == HICOM_DT_F32, (_Float16*)dst, (long)n);
    return hicom_host::check_launch("to_f16");
}

// Checks one problem (+ role) and fills the kernel's parameter block.  `force_aux` > 0: the role runs on exactly that many workgroups.
static int r16_build(const void* a, const void* w, const void* b, int32_t b_dt,
                     int32_t M, int32_t N, int32_t K, int32_t act, void* out_f16,
                     void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group,
                     const hicom_r16_role* role, int force_aux, R16Params& p, int& n_aux) {
    const bool role_only = !a && !w && M == 0;                    // (hicom_gemv_chain_fwd: no tile grid, every workgroup runs the role)
    if (!role_only) {
        HICOM_REQUIRE(a && w, HICOM_EINVAL, "readout16_gemm: NULL pointer");
        HICOM_REQUIRE(out_f16 || y, HICOM_EINVAL, "readout16_gemm: no output");
        HICOM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, HICOM_EINVAL, "readout16_gemm: bad shape M=%d N=%d K=%d (K %% 64)", M, N, K);
    }
    HICOM_REQUIRE(!y || (ldy >= N && row0 >= 0 && nl_group >= 0), HICOM_EINVAL, "readout16_gemm: bad output layout");
    HICOM_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)w % 16 == 0) && M < (1 << 30), HICOM_EINVAL, "readout16_gemm: alignment");
    const bool vec = N % 4 == 0 && (!out_f16 || ((uintptr_t)out_f16 % 8 == 0)) && (!y || (ldy % 4 == 0 && (uintptr_t)y % 16 == 0));
    p.a = (const _Float16*)a; p.w = (const _Float16*)w; p.b = b; p.b_f32 = b_dt == HICOM_DT_F32;
    p.M = M; p.N = N; p.K = K; p.act = act; p.o16 = (_Float16*)out_f16; p.y = y; p.y_f32 = y_dt == HICOM_DT_F32;
    p.ldy = (long)ldy; p.row0 = (long)row0; p.nl_group = nl_group; p.vec = vec ? 1 : 0; p.bvec = (b && (uintptr_t)b % 16 == 0) ? 1 : 0;
    {
        const bool y16 = y && y_dt != HICOM_DT_F32;
        p.line16 = (N % 8 == 0 && ((out_f16 && !y && (uintptr_t)out_f16 % 16 == 0) || (!out_f16 && y16 && (uintptr_t)y % 16 == 0 && ldy % 8 == 0))) ? 1 : 0;
    }
    // tile shape: 96 x 64, or -- wide layers whose grid is more ROUNDS of one-per-CU workgroups than a larger tile's -- 96 x 128, or 192 x 128
    // (HICOM_R16_TN = 64 | 128 | 192x128 forces one: A/B switch).  The smallest tile among those with the fewest rounds.
    int tn = kRN, tm = kRM;
    if (!role_only && N % 128 == 0) {
        static int tn_env = -1;
        if (tn_env < 0) {
            const char* e = getenv("HICOM_R16_TN");
            tn_env = (e && !strcmp(e, "128")) ? 128 : (e && !strcmp(e, "64")) ? 64 : (e && !strcmp(e, "192x128")) ? 192 : 0;
        }
        auto rounds = [&](int tm_, int tn_) { return (((M + tm_ - 1) / tm_) * (N / tn_) + 255) / 256; };
        const int r64 = rounds(96, 64), r128 = rounds(96, 128), r192 = rounds(192, 128);
        if (tn_env == 128 || (tn_env == 0 && r128 < r64)) tn = 128;
        if (tn_env == 192 || (tn_env == 0 && r192 < r128 && r192 < r64)) {
            tn = 128;
            tm = 192;
        }
    }
    p.tn = tn;
    p.tm = tm;
    const int nby = (M + tm - 1) / tm;
    const int nbx = (N + tn - 1) / tn;
    p.n_gemm = role_only ? 0 : 8 * ((nbx * nby + 7) / 8);
    n_aux = 0;
    const AuxGemv none{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, 0, 0, 0, 0, nullptr, 0};
    p.aux = none;
    p.aux2 = none;
    p.role = HICOM_ROLE_NONE;
    p.chain_state = nullptr;
    p.mv = MergeVprojFixParams{nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr};
    p.cg_cpw1 = p.cg_cpw2 = p.cg_r1 = p.cg_r2 = 0;
    auto conv = [](const hicom_aux_gemv* aux) {
        return AuxGemv{aux->xs, aux->x_parts, (long)aux->x_stride, (const uint16_t*)aux->xb, aux->w,
                       aux->b, (const uint16_t*)aux->res, aux->N, aux->K, aux->act, aux->y, aux->w_dt == HICOM_DT_F32, aux->b_dt == HICOM_DT_F32,
                       aux->rows_dst, aux->rows_dt == HICOM_DT_F32, aux->rows_dst ? aux->rows_reps : 0, (long)aux->rows_ld, (long)aux->rows_row0,
                       (const long long*)aux->x_fixed, aux->x_fixed_clear ? 1 : 0};
    };
    auto check_gemv = [](const hicom_aux_gemv* aux, bool needs_x, bool needs_out = true, int kmax = 1536) -> int {
        HICOM_REQUIRE(aux->w && (!needs_out || aux->y || aux->rows_dst) && aux->N > 0 && aux->K > 0 && aux->K % 8 == 0 && aux->K <= kmax && ((uintptr_t)aux->w % 16 == 0), HICOM_EINVAL,
                      "readout16_gemm: aux GEMV arguments");
        if (aux->x_fixed) HICOM_REQUIRE((uintptr_t)aux->x_fixed % 8 == 0, HICOM_EINVAL, "readout16_gemm: aux x_fixed alignment");
        else if (needs_x) HICOM_REQUIRE(aux->xs && aux->x_parts > 0 && aux->x_stride % 4 == 0 && ((uintptr_t)aux->xs % 16 == 0) && (long)aux->x_parts * aux->K <= 28 * 1024 &&
                                            (1536 + (long)aux->x_parts * aux->K) * 4 <= 6 * kRStage, HICOM_EINVAL, "readout16_gemm: aux GEMV partial vectors");
        HICOM_REQUIRE(!aux->rows_dst || (aux->rows_reps > 0 && aux->rows_ld >= aux->N && aux->rows_row0 >= 0), HICOM_EINVAL, "readout16_gemm: aux row destination");
        return HICOM_OK;
    };
    const int kind = role ? role->kind : HICOM_ROLE_NONE;
    if (kind == HICOM_ROLE_GEMV && role->gemv.N > 0) {
        if (int rc = check_gemv(&role->gemv, true)) return rc;
        p.aux = conv(&role->gemv);
        p.role = HICOM_ROLE_GEMV;
    } else if (kind == HICOM_ROLE_GEMV_CHAIN) {
        if (int rc = check_gemv(&role->gemv, true, false)) return rc;          // (the first layer's result travels as granules: y optional)
        if (int rc = check_gemv(&role->gemv2, false, true, kChainBigN)) return rc;      // (the second layer's K is the first layer's width: up to 4096)
        HICOM_REQUIRE(role->gemv.N <= kChainBigN, HICOM_EUNSUP, "readout16_gemm: GEMV chain: first layer of %d columns (<= %d)", role->gemv.N, kChainBigN);
        HICOM_REQUIRE(role->gemv.x_fixed && role->gemv2.K == role->gemv.N && role->gemv2.w_dt == HICOM_DT_BF16 && role->chain_state &&
                          (uintptr_t)role->chain_state % 16 == 0, HICOM_EINVAL,
                      "readout16_gemm: GEMV chain (first layer from x_fixed, second layer bf16 with K = the first layer's N, state block)");
        HICOM_REQUIRE(role->gemv.K % 8 == 0 && role->gemv.N % 8 == 0 && ((uintptr_t)role->gemv2.w % 16 == 0), HICOM_EINVAL,
                      "readout16_gemm: GEMV chain: K and N of the first layer must be multiples of 8 (16-byte weight rows)");
        p.aux = conv(&role->gemv);
        p.aux2 = conv(&role->gemv2);
        p.chain_state = (unsigned*)role->chain_state;
        p.role = HICOM_ROLE_GEMV_CHAIN;
    } else if (kind == HICOM_ROLE_MERGE_VPROJ) {
        HICOM_REQUIRE(role->part_m && role->part_l && role->part_acc && (!role->w_v == !role->o_fix) && (role->w_v || (role->out_ml && role->out_ctx)), HICOM_EINVAL,
                      "readout16_gemm: merge role: NULL pointer (w_v and o_fix go together; without them out_ml and out_ctx are the result)");
        HICOM_REQUIRE(role->part_dt == HICOM_DT_F16 && role->nparts > 0 && role->nparts <= 256 && role->rows > 0 && role->rows <= role->rows_pad &&
                          role->E > 0 && role->E % 64 == 0 && role->E % role->rows == 0 && role->E / role->rows <= 128 &&
                          ((uintptr_t)role->o_fix % 8 == 0) && ((uintptr_t)role->w_v % 16 == 0) && ((uintptr_t)role->part_acc % 16 == 0),
                      HICOM_EINVAL, "readout16_gemm: merge role: fp16 partial contexts, nparts <= 256, head dim <= 128, E %% 64, alignment");
        p.mv = MergeVprojFixParams{role->part_m, role->part_l, role->part_acc, role->nparts, role->rows_pad, role->E, role->E / role->rows,
                                   (const uint16_t*)role->w_v, (long long*)role->o_fix, role->out_ml, role->out_ctx};
        p.mv.ctx_unnorm = role->ctx_unnorm ? 1 : 0;
        if (role->part_marg) {
            // value-side pos-emb in the merge (merge_item.hpp): marginals of the partials + the weight-only table v_proj . pe^T
            HICOM_REQUIRE(role->vpe_f16 && role->w_v && role->marg_slots == 8 * (role->E / 64) && (uintptr_t)role->part_marg % 16 == 0 &&
                              (uintptr_t)role->vpe_f16 % 16 == 0, HICOM_EINVAL,
                          "readout16_gemm: merge role: part_marg goes with vpe_f16 and w_v, marg_slots = 8 * (E / 64) = %d, 16-byte alignment", 8 * (role->E / 64));
            p.mv.part_marg = (const _Float16*)role->part_marg;
            p.mv.vpe16 = (const _Float16*)role->vpe_f16;
            p.mv.marg_slots = role->marg_slots;
        }
        p.role = HICOM_ROLE_MERGE_VPROJ;
    } else {
        HICOM_REQUIRE(kind == HICOM_ROLE_NONE || kind == HICOM_ROLE_GEMV, HICOM_EINVAL, "readout16_gemm: role kind %d", kind);
    }
    if (p.role != HICOM_ROLE_NONE) {
        // the CUs the tile grid leaves idle (one workgroup per CU: the ring takes 160 KB of LDS), at least 16
        // the same number on every XCD, and ONE CU per XCD left free: with every CU of an XCD spoken for (25 tiles + 7 aux
        // = 32) an aux workgroup was seen queueing behind a 12-us tile (GEMM 1 in situ: 16.6 us with aux, 12.2 without)
        n_aux = ((256 - p.n_gemm) / 8 - 1) * 8;
        if (n_aux < 16) n_aux = 16;
        if (n_aux > 72) n_aux = 72;
        // the two-layer chain is bound by the bytes a role CU pulls (4.1 + 1.6 MB of weights beside the streaming tiles): every CU the
        // tile grid leaves free takes a share (the four tile slots beyond the 196 tiles exit at once)
        if (p.role == HICOM_ROLE_GEMV_CHAIN && 256 - p.n_gemm >= 16) n_aux = 256 - p.n_gemm < 72 ? 256 - p.n_gemm : 72;
        // a launch of its own (the FINISH phase of the frame-sharded step, on the comm stream): it runs BESIDE the next step's ring kernel,
        // which holds 216 of the 256 CUs with 160 KB of LDS each -- 32 role workgroups (one per CU: the role takes the whole LDS too) all
        // find a CU among the 40 it leaves free.  With 64, two dozen waited for the ring kernel to end while the resident ones spun on their
        // hand-off, and then competed with readout GEMM 1 for CUs: the pipelined step read 93.8 us against 76 for the plain one.  A fixed
        // count per state block (the epoch arithmetic of the hand-off).
        if (role_only) n_aux = 32;
        if (p.role == HICOM_ROLE_MERGE_VPROJ) {
            // every item of a role workgroup in flight at once: ceil(items / kMvRoleItems) workgroups (162 items -> 54), never more than
            // the CUs the tile grid leaves free (one workgroup per CU: the launch asks for 160 KB of LDS)
            const int items = (p.mv.E / 64) * (p.mv.E / p.mv.hd);
            const int want = (items + kMvRoleItems - 1) / kMvRoleItems;
            if (want > n_aux && want <= 256 - p.n_gemm - 2) n_aux = want;
        }
        if (force_aux > 0) n_aux = force_aux;
    }
    if (p.role == HICOM_ROLE_GEMV_CHAIN) {
        const ChainGeom cg = chain_geom(p.aux.N, p.aux.K, p.aux.w_f32 != 0, p.aux2.N, n_aux);
        HICOM_REQUIRE(cg.r1 >= 1 && cg.r2 >= 1 && cg.cpw1 <= 256 && cg.cpw2 <= 256, HICOM_EUNSUP,
                      "readout16_gemm: GEMV chain: layers of %d and %d columns over %d role workgroups do not fit", p.aux.N, p.aux2.N, n_aux);
        p.cg_cpw1 = cg.cpw1; p.cg_cpw2 = cg.cpw2; p.cg_r1 = cg.r1; p.cg_r2 = cg.r2;
    }
    return HICOM_OK;
}

static int readout16_launch(const void* a, const void* w, const void* b, int32_t b_dt,
                            int32_t M, int32_t N, int32_t K, int32_t act, void* out_f16,
                            void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group,
                            const hicom_r16_role* role, void* stream) {
    R16Params p;
    int n_aux = 0;
    if (int rc = r16_build(a, w, b, b_dt, M, N, K, act, out_f16, y, y_dt, ldy, row0, nl_group, role, 0, p, n_aux)) return rc;
    // 8 ring stages (160 KB: one workgroup per CU); 6 stages measured 0.5 us slower per GEMM (tools/gpu_round_c.sh, round 2)
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(readout16_gemm_kernel<8, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kRStage);
        hipFuncSetAttribute(reinterpret_cast<const void*>(readout16_gemm_kernel<8, 64, 96, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kRStage);
        hipFuncSetAttribute(reinterpret_cast<const void*>(readout16_gemm_kernel<5, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kRStage);
        hipFuncSetAttribute(reinterpret_cast<const void*>(readout16_gemm_kernel<4, 128, 192>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kRStage);
        attr_set = true;
    }
    // role workgroups first in dispatch order would delay tiles on their CUs; they go last and land on the free CUs
    // (both forms ask for the whole 160 KB: one workgroup per CU, and the roles use it)
    const dim3 grid((unsigned)(p.n_gemm + n_aux));
#define R16_ARGS0 (const void*)p.a, (const void*)p.w, (const void*)nullptr, (const void*)nullptr, p.M, p.N, p.K, p.n_gemm, p.role, n_aux, p
    if (p.tm == 192) HICOM_LAUNCH((readout16_gemm_kernel<4, 128, 192>), grid, dim3(256), 8 * kRStage, (hipStream_t)stream, R16_ARGS0);
    else if (p.tn == 128) HICOM_LAUNCH((readout16_gemm_kernel<5, 128>), grid, dim3(256), 8 * kRStage, (hipStream_t)stream, R16_ARGS0);
    // (HICOM_R16_PRE=0: dev A/B switch -- the chain launch with the tiles' arguments preloaded, like every other launch)
    static const bool pre_role = !(getenv("HICOM_R16_PRE") && getenv("HICOM_R16_PRE")[0] == '0');
    if (p.tm == 192 || p.tn == 128) {
    } else if (pre_role && p.role == HICOM_ROLE_GEMV_CHAIN && p.n_gemm >= 0 && p.n_gemm < 65536 && p.aux.N < 65536 && p.aux2.N < 65536)
        HICOM_LAUNCH((readout16_gemm_kernel<8, 64, 96, 1>), grid, dim3(256), 8 * kRStage, (hipStream_t)stream, (const void*)p.chain_state, (const void*)p.aux.x_fixed,
                     (const void*)p.aux.w, (const void*)p.aux2.w, p.aux.K, p.aux.N | (p.aux2.N << 16), p.cg_cpw1 | (p.cg_cpw2 << 16), p.n_gemm | (n_aux << 16),
                     p.role | ((p.aux.w_f32 ? 1 : 0) << 8), p.cg_r1 | (p.cg_r2 << 16), p);
    else HICOM_LAUNCH((readout16_gemm_kernel<8, 64>), grid, dim3(256), 8 * kRStage, (hipStream_t)stream, R16_ARGS0);
#undef R16_ARGS0
    return hicom_host::check_launch("readout16_gemm");
}

extern "C" int hicom_readout16_gemm_fwd(const void* a, const void* w, const void* b, int32_t b_dt,
                                        int32_t M, int32_t N, int32_t K, int32_t act, void* out_f16,
                                        void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group,
                                        const hicom_aux_gemv* aux, void* stream) {
    if (!aux || aux->N <= 0) return readout16_launch(a, w, b, b_dt, M, N, K, act, out_f16, y, y_dt, ldy, row0, nl_group, nullptr, stream);
    hicom_r16_role role;
    memset(&role, 0, sizeof(role));
    role.kind = HICOM_ROLE_GEMV;
    role.gemv = *aux;
    return readout16_launch(a, w, b, b_dt, M, N, K, act, out_f16, y, y_dt, ldy, row0, nl_group, &role, stream);
}

extern "C" int hicom_readout16_gemm_role_fwd(const void* a, const void* w, const void* b, int32_t b_dt,
                                             int32_t M, int32_t N, int32_t K, int32_t act, void* out_f16,
                                             void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group,
                                             const hicom_r16_role* role, void* stream) {
    return readout16_launch(a, w, b, b_dt, M, N, K, act, out_f16, y, y_dt, ldy, row0, nl_group, role, stream);
}

#ifdef HICOM_TRACE
extern "C" int hicom_debug_r16_trace(void* dst, int64_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(hicom::g_r16_trace), (size_t)bytes) == hipSuccess ? HICOM_OK : HICOM_ELAUNCH;
}
#endif

extern "C" int hicom_gemv_chain_fwd(const hicom_r16_role* role, void* stream) {
    HICOM_REQUIRE(role && role->kind == HICOM_ROLE_GEMV_CHAIN, HICOM_EINVAL, "gemv_chain: a HICOM_ROLE_GEMV_CHAIN role");
    return readout16_launch(nullptr, nullptr, nullptr, 0, 0, 64, 64, HICOM_ACT_NONE, nullptr, nullptr, 0, 0, 0, 0, role, stream);
}

// ---- fused tail launch (see readout_tail_kernel) ----
constexpr int kTailRole = 54;                 // role workgroups: ceil(162 merge items / 3) at the release shape; 200 + 54 = 254 of 256 CUs
constexpr int kTailSyncLines = 2 + 64;        // arrivals, gate, up to 64 row blocks
constexpr int kTailFlags = 256;               // one ownership flag per GEMM-2 tile

extern "C" int64_t hicom_readout_tail_state_bytes(void) { return (int64_t)kTailSyncLines * 128 + kTailFlags * 8; }

extern "C" int hicom_readout_tail_fwd(const hicom_r16_gemm* g1, const hicom_r16_gemm* g2, const hicom_r16_role* merge,
                                      const hicom_r16_role* chain, void* state, void* stream) {
    HICOM_REQUIRE(g1 && g2 && merge && chain && state && (uintptr_t)state % 128 == 0, HICOM_EINVAL, "readout_tail: NULL pointer / state alignment");
    HICOM_REQUIRE(merge->kind == HICOM_ROLE_MERGE_VPROJ && chain->kind == HICOM_ROLE_GEMV_CHAIN, HICOM_EINVAL, "readout_tail: a merge role and a chain role");
    HICOM_REQUIRE(g1->out_f16 && !g1->y && g2->a == g1->out_f16 && g2->M == g1->M && g2->K == g1->N && !g2->out_f16 && g2->y, HICOM_EINVAL,
                  "readout_tail: GEMM 2 reads GEMM 1's fp16 plane");
    TailParams q;
    int na1 = 0, na2 = 0;
    if (int rc = r16_build(g1->a, g1->w, g1->b, g1->b_dt, g1->M, g1->N, g1->K, g1->act, g1->out_f16, nullptr, 0, 0, 0, 0, merge, kTailRole, q.g1, na1)) return rc;
    if (int rc = r16_build(g2->a, g2->w, g2->b, g2->b_dt, g2->M, g2->N, g2->K, g2->act, nullptr, g2->y, g2->y_dt, g2->ldy, g2->row0, g2->nl_group, chain,
                           kTailRole, q.g2, na2)) return rc;
    const int nby = (g1->M + kRM - 1) / kRM, nbx = g1->N / kRN, items = (q.g1.mv.E / 64) * (q.g1.mv.E / q.g1.mv.hd);
    HICOM_REQUIRE(q.g1.line16 && q.g2.line16 && q.g1.tn == kRN && q.g2.tn == kRN && q.g1.tm == kRM && q.g2.tm == kRM && g1->N % kRN == 0 && g2->N == g1->N && g2->K / 64 >= 8 + 6 && nby + 2 <= kTailSyncLines &&
                      nbx * nby <= kTailFlags && q.g1.n_gemm + kTailRole <= 256 && items <= kMvRoleItems * kTailRole && q.g1.mv.wv,
                  HICOM_EUNSUP, "readout_tail: shape outside the fused form (two layers of the same width in whole 64-column tiles, K2 >= 896, "
                                "<= 202 tiles, <= 162 merge items)");
    q.sy.arrivals = (unsigned long long*)state;
    q.sy.gate = (unsigned long long*)state + 16;
    q.sy.rowblk = (unsigned long long*)state + 32;
    q.sy.flags = (unsigned long long*)state + 16 * kTailSyncLines;
    q.sy.per_epoch = nbx;
    q.n1 = q.g1.n_gemm;
    q.n_role = kTailRole;
    q.tiles = nbx * nby;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(readout_tail_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kRStage);
        attr_set = true;
    }
    HICOM_LAUNCH(readout_tail_kernel<8>, dim3((unsigned)(q.n1 + q.n_role)), dim3(256), 8 * kRStage, (hipStream_t)stream, q);
    return hicom_host::check_launch("readout_tail");
}

extern "C" int64_t hicom_r16_chain_state_bytes(int32_t n_mid) { return n_mid > 0 ? (int64_t)n_mid * 8 + kChainStateHead : HICOM_EINVAL; }
