// Global compressor: streaming multi-query cross-attention over ALL visual tokens, K = V = x.
//
// Replaces MultiheadAttention.forward (reference projector.py:166-228) as used by
// GlobalCompressor.forward (:634-646).  k_proj / v_proj are folded into the queries on the
// host side of the ABI (hicom_fold_query_fwd), so this kernel consumes the RAW bf16 tokens:
//     S[r, n]   = qt[r, :] . x[n, :]  (+ separable positional terms a_t + a_y + a_x)
//     ACC[r, :] = sum_n exp(S[r, n] - m_r) x[n, :]          (online softmax, fp32)
// and never materialises K, V or x + pos.  x is read from HBM exactly once per row group.
//
// CDNA4 mapping
//   * workgroup = 4 waves, 2 workgroups per CU (80 KiB LDS, <=256 VGPR each); a workgroup walks
//     a contiguous range of 16-token tiles with running (m, l, ACC) in registers.
//   * tile staging: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction, whole 256-B
//     row segments) into a double-buffered [E/128][16 rows][256 B] image with the
//     chunk ^ rot(row) swizzle, so that BOTH the row read (ds_read_b128: B operand of the
//     score MFMA) and the transposed read (ds_read_b64_tr_b16: B operand of the P.x MFMA)
//     of the same image are bank-conflict free / 2-way.
//   * scores: v_mfma_f32_16x16x32_bf16, A = folded queries as bf16 hi + lo pairs (two MFMAs
//     into one accumulator -> fp32-grade logits), each wave contracts its own E/4 channel
//     slice and the four partial 16x16 tiles are summed through LDS.
//   * softmax fused at wavefront level: lane (row, token-quad) owns 4 logits, row max / row sum
//     are two cross-lane shuffles; P is split hi/lo and is already in A-fragment order.
//   * P.x: v_mfma_f32_16x16x16_bf16, each wave owns E/4 output channels (72 accumulator VGPRs).
#include <stdlib.h>

#include "common.hpp"

namespace hicom {

struct StreamParams {
    const uint16_t* x;
    long N;
    const uint16_t* qhi;
    const uint16_t* qlo;
    const float* pos_a;
    int pos_stride;
    int H, W, HW;
    int t0i, y0i, x0i;
    float* scores;
    long score_stride;
    float* part_m;
    float* part_l;
    float* part_acc;
    int rows_pad;
    int rows;      // rows < rows_pad that carry real queries; padded rows are never stored
    int ntiles;
    // backward mode (global_stream_kernel<NB, true>): softmax state and scores of the forward pass
    const float* s_in;     // [rows_pad][score_stride] raw logits S written by the forward pass
    const float* ml;       // [rows][2] (M, L) of the forward softmax
    const float* delta;    // [rows] dctx_r . ctx_r
    // clip-scale variant (forward): logit = (qt . x + pos + row_const[r]) * inv_norm[n]
    const float* inv_norm; // [N] or NULL
    const float* row_const;// [rows_pad] or NULL
    // wide forward only: un-normalised positional marginals of the softmax weights of each token chunk, relative to the chunk's
    // running max like part_acc: [nparts][rows_pad][marg_stride] = [kWideFrames frames from the chunk's first | pad to 16 |
    // H grid rows | pad to 16 | W grid columns | pad to 16], or NULL (then `scores` must be given: the marginals are taken from it)
    float* part_marg;
    int marg_stride;
};

// bit rotation of the row index used as the 16-byte-chunk swizzle (bijective on 0..15)
__device__ __forceinline__ int swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
// token-quad permutation: MFMA k-group g <-> tile rows 4*sig(g)..4*sig(g)+3, so that the two
// 4-row blocks a 32-lane half reads with ds_read_b64_tr_b16 sit 8 rows apart (conflict-free)
__device__ __forceinline__ int sig(int g) { return ((g & 1) << 1) | (g >> 1); }

// BWD = false: forward (above).  BWD = true: attention backward over the same stream (reference autograd through
// projector.py:197-215): the "queries" are the upstream gradients dctx_r of the per-head contexts, so the MFMA score
// tile is dP[r, n] = dctx_r . (x_n + pos_n); with the forward's logits S and softmax state (M, L) the lane forms
//     dS[r, n] = exp(S[r, n] - M_r) / L_r * (dP[r, n] - delta_r),      delta_r = dctx_r . ctx_r
// writes it to `scores` (the positional marginals are taken from it afterwards) and accumulates
//     ACC[r, :] += dS[r, n] x[n, :]        ( = the x part of d q~_r )
// through the same hi/lo P.x MFMAs.  No running max: the state is known.
template <int NB, bool BWD = false>
__global__ __launch_bounds__(256, 2) void global_stream_kernel(StreamParams p) {
    constexpr int E = NB * 128;
    constexpr int SLICE = E / 4;          // channels owned by one wave
    constexpr int KSTEPS = SLICE / 32;    // score MFMAs (K = 32 channels) per wave per pass
    constexpr int CBLK = SLICE / 16;      // 16-channel output blocks per wave
    constexpr int TILE_BYTES = NB * 4096;
    constexpr int PIECES = NB * 4;        // 1 KiB LDS-DMA pieces per tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tilebuf = smem;                                           // [2][TILE_BYTES]
    float* red = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);  // [4 waves][16 rows][16 tokens]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const int part = blockIdx.x, rg = blockIdx.y, nparts = gridDim.x;
    const int tb = (int)(((long)p.ntiles * part) / nparts);
    const int te = (int)(((long)p.ntiles * (part + 1)) / nparts);
    const long row_glob = (long)rg * 16 + r16;

    // ---- A operand: folded queries, this wave's channel slice, hi and lo parts ----------------
    bf16x8 ahi[KSTEPS], alo[KSTEPS];
    {
        const long off = row_glob * E + SLICE * wave + 8 * kg;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            ahi[s] = *reinterpret_cast<const bf16x8*>(p.qhi + off + 32 * s);
            alo[s] = *reinterpret_cast<const bf16x8*>(p.qlo + off + 32 * s);
        }
    }

    f32x4 acc[CBLK];
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;   // online-softmax state of row r16 (replicated over kg, waves)

    // ---- LDS-DMA staging of one 16-token tile ------------------------------------------------
    auto stage = [&](int tile, int buf) {
        const int r = lane >> 4, cpos = lane & 15;
        static_assert(PIECES % 4 == 0, "pieces are dealt round-robin to the 4 waves");
#pragma unroll
        for (int i = 0; i < PIECES / 4; ++i) {
            const int pi = wave + 4 * i;
            const int blk = pi >> 2, row = 4 * (pi & 3) + r;
            long tok = (long)tile * 16 + row;
            tok = tok < p.N ? tok : p.N - 1;   // tail tile: re-read a valid row (masked later)
            const char* src = reinterpret_cast<const char*>(p.x) + tok * (long)(E * 2) + blk * 256 +
                              16 * (cpos ^ swz(row));
            char* dst = tilebuf + buf * TILE_BYTES + pi * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                             (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
        }
    };

    // per-lane LDS byte offsets inside a tile image
    const int q4 = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 4 * sig(kg) + q4;                // row this lane addresses in transposed reads
    const int rd_row_off = r16 * 256, rd_swz = swz(r16);
    const int tr_row_off = trow * 256 + 8 * (pp & 1), tr_swz = swz(trow);
    const float* pa = p.pos_a ? p.pos_a + row_glob * p.pos_stride : nullptr;

    // positional logit terms are fetched ONE TILE AHEAD: vmcnt retires in order, so consuming an
    // ordinary load issued after the LDS-DMA of the next tile would drain that prefetch.
    float pt[4] = {0.f, 0.f, 0.f, 0.f}, py[4] = {0.f, 0.f, 0.f, 0.f}, px[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch_pos = [&](int tile) {
        if (pa) {
            const long nb = (long)tile * 16 + 4 * sig(kg);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                long n = nb + j;
                n = n < p.N ? n : p.N - 1;
                const unsigned un = (unsigned)n;
                const unsigned t = un / (unsigned)p.HW, rem = un - t * (unsigned)p.HW;
                const unsigned y = rem / (unsigned)p.W, xx = rem - y * (unsigned)p.W;
                pt[j] = pa[p.t0i + t];
                py[j] = pa[p.y0i + y];
                px[j] = pa[p.x0i + xx];
            }
        }
    };

    // backward: the forward's logits of (row r16, tokens 4*sig(kg)..+3), fetched one tile ahead like the positional terms
    f32x4 s_next = f32x4{0.f, 0.f, 0.f, 0.f};
    float bw_m = 0.f, bw_linv = 0.f, bw_delta = 0.f;
    const float crow = (!BWD && p.row_const) ? p.row_const[row_glob] : 0.f;
    auto fetch_s = [&](int tile) {
        if constexpr (BWD) {
            s_next = *reinterpret_cast<const f32x4*>(p.s_in + row_glob * p.score_stride + (long)tile * 16 + 4 * sig(kg));
        } else if (p.inv_norm) {        // per-token key norms of this lane's 4 tokens (tail tile: clamped, masked later)
            const long nb = (long)tile * 16 + 4 * sig(kg);
#pragma unroll
            for (int j = 0; j < 4; ++j) s_next[j] = p.inv_norm[nb + j < p.N ? nb + j : p.N - 1];
        }
    };
    if constexpr (BWD) {
        if (row_glob < p.rows) {
            bw_m = p.ml[2 * row_glob];
            bw_linv = 1.0f / p.ml[2 * row_glob + 1];
            bw_delta = p.delta[row_glob];
        }
    }

    if (tb < te) {
        fetch_pos(tb);
        fetch_s(tb);
        stage(tb, 0);
    }

    for (int tile = tb; tile < te; ++tile) {
        const int cur = (tile - tb) & 1;
        // [A] every wave's LDS-DMA pieces of tile `tile` have landed (explicit drain: hipcc does
        //     not order LDS-DMA against the barrier by itself), and every wave is done reading the
        //     other buffer and the reduction scratch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float padd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) padd[j] = pt[j] + py[j] + px[j];
        const f32x4 s_cur = s_next;
        if (tile + 1 < te) {
            fetch_pos(tile + 1);
            fetch_s(tile + 1);
            stage(tile + 1, cur ^ 1);
        }
        const char* img = tilebuf + cur * TILE_BYTES;

        // ---- partial scores over this wave's channel slice: S = qt_hi.x^T + qt_lo.x^T -------
        f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int ch0 = SLICE * wave + 32 * s;
            const int blk = ch0 >> 7, cbase = (ch0 & 127) >> 3;
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(img + blk * 4096 + rd_row_off +
                                                              16 * ((cbase + kg) ^ rd_swz));
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[s], b, s4, 0, 0, 0);
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[s], b, s4, 0, 0, 0);
        }
        // C layout: lane holds rows 4*kg + j (j = 0..3) of token column r16
        float* rw = red + wave * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) rw[(4 * kg + j) * 16 + r16] = s4[j];
        lds_barrier();   // [B] LDS only: the next tile's LDS-DMA stays in flight

        // ---- full logits of (row r16, tokens 4*sig(kg) .. +3) -----------------------------------
        const float* rb = red + r16 * 16 + 4 * sig(kg);
        f32x4 lg = *reinterpret_cast<const f32x4*>(rb);
        lg += *reinterpret_cast<const f32x4*>(rb + 256);
        lg += *reinterpret_cast<const f32x4*>(rb + 512);
        lg += *reinterpret_cast<const f32x4*>(rb + 768);
        const long n0 = (long)tile * 16 + 4 * sig(kg);
#pragma unroll
        for (int j = 0; j < 4; ++j) lg[j] += padd[j];
        if (!BWD && p.inv_norm) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lg[j] = (lg[j] + crow) * s_cur[j];
        }
        float pr[4];
        if constexpr (BWD) {
            // dS = softmax weight x (dP - delta); padded rows (state 0, 0, 0) and tail tokens contribute nothing
#pragma unroll
            for (int j = 0; j < 4; ++j)
                pr[j] = (n0 + j < p.N && row_glob < p.rows) ? expf(s_cur[j] - bw_m) * bw_linv * (lg[j] - bw_delta) : 0.f;
            if (wave == 0) *reinterpret_cast<f32x4*>(p.scores + row_glob * p.score_stride + n0) = f32x4{pr[0], pr[1], pr[2], pr[3]};
        } else {
        if (wave == 0) *reinterpret_cast<f32x4*>(p.scores + row_glob * p.score_stride + n0) = lg;

        // ---- online softmax for row r16 (lanes r16, r16+16, r16+32, r16+48 share the row) -------
        float tmax = -1.0e30f;
#pragma unroll
        for (int j = 0; j < 4; ++j) tmax = (n0 + j < p.N) ? fmaxf(tmax, lg[j]) : tmax;
        tmax = xrow4_max(tmax);                          // (the row's four lanes r16 + 16 kg: gfx950 row swaps, no LDS round trip)
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pr[j] = (n0 + j < p.N) ? expf(lg[j] - m_new) : 0.f;
            lsum += pr[j];
        }
        lsum = xrow4_sum(lsum);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        // rescale the running context when any row's max moved (wave-uniform branch)
        if (__any(alpha != 1.0f)) {
            const float a0 = __shfl(alpha, 4 * kg + 0, 64), a1 = __shfl(alpha, 4 * kg + 1, 64);
            const float a2 = __shfl(alpha, 4 * kg + 2, 64), a3 = __shfl(alpha, 4 * kg + 3, 64);
#pragma unroll
            for (int cb = 0; cb < CBLK; ++cb) {
                acc[cb][0] *= a0; acc[cb][1] *= a1; acc[cb][2] *= a2; acc[cb][3] *= a3;
            }
        }
        }
        bf16x4 phi, plo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint16_t h, l;
            split_bf16(pr[j], h, l);
            phi[j] = (short)h;
            plo[j] = (short)l;
        }

        // ---- ACC += P . x over this wave's output channels ----------------------------------------
#pragma unroll
        for (int cb = 0; cb < CBLK; ++cb) {
            const int ch0 = SLICE * wave + 16 * cb;
            const int blk = ch0 >> 7, c2 = (ch0 & 127) >> 3;
            const char* a = img + blk * 4096 + tr_row_off + 16 * ((c2 + (pp >> 1)) ^ tr_swz);
            const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(phi, b, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(plo, b, acc[cb], 0, 0, 0);
        }
    }

    // ---- partial results of this token chunk ----------------------------------------------------
    const long prow = (long)part * p.rows_pad + rg * 16;
    if (!BWD && wave == 0 && kg == 0 && rg * 16 + r16 < p.rows) {
        p.part_m[prow + r16] = m_run;
        p.part_l[prow + r16] = l_run;
    }
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) {
        float* o = p.part_acc + (prow + 4 * kg) * E + SLICE * wave + 16 * cb + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rg * 16 + 4 * kg + j < p.rows) o[(long)j * E] = acc[cb][j];
    }
}

// Dev-only phase sums of the many-row kernel (tools/wide_trace.py builds a second library with -DHICOM_WTRACE): lane 0 of every wave
// adds up the cycles between ten points of its tile iteration; [block][wave][phase].
#ifdef HICOM_WTRACE
__device__ unsigned long long g_wide_trace[512 * 8 * 12];
#define HICOM_WT(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); wt_sum[i] += now_ - wt_last; wt_last = now_; } while (0)
#else
#define HICOM_WT(i) do {} while (0)
#endif

constexpr int kWideRG = 2;
constexpr int kWideFrames = 8;      // frames a workgroup's token range may touch
constexpr int kWideMargBlocks = 2;  // 16-column marginal blocks per wave (x 4 waves of a row group)

// ---------------------------------------------------------------------------------------------
// Wide form for many query rows (guide off / coarse / fine: 32 queries x 9 heads = 288 folded rows).  The narrow kernel above gives
// every 16-row group its own pass over the token stream (18 passes = 1.9 GB through the memory system).  Here a 512-thread
// workgroup (one per CU) owns RG = 2 row groups: waves 0-3 and waves 4-7 are two copies of the 4-wave machine above working on
// the SAME LDS tile, so the stream is read half as often; three-tile ring (waves 0-3 issue the DMA), the positional tables of the
// 32 rows in LDS (frames this workgroup's tokens can touch | rows | columns) with the tile's token coordinates tabled once per
// tile by an otherwise idle lane group, P.x with one K = 32 MFMA per 16-channel block (P hi | lo against the fragment twice).
// Software-pipelined (round 3): the round-2 form spent a tile as a serial chain
//   [A] -> score MFMAs -> partial store -> [B] -> exchange / positional lookups / softmax (VALU) -> P.x MFMAs -> [A]
// with every wave of the workgroup in lock-step (262 us alone at C2).  Here an iteration holds the work of TWO tiles that do not
// depend on each other: the score MFMAs of tile i + 1 are issued first and run in the matrix pipe while the same wave's VALU does
// the exchange / softmax of tile i, then P.x of tile i; the partial logits of tile i + 1 go to the OTHER half of a double-
// buffered exchange area at the end.  ONE barrier per tile: it publishes the partials of tile i + 1 and the DMA'd tile i + 2
// (the loading waves wait for their pieces right before it) and retires tile i's buffer, which the next iteration restages.
// Same ring depth (3 tiles: P.x | scores | in flight), same results bit for bit (same operations in the same order per row):
// 239 us alone, guide-off forward 414 -> 392 us.  A four-buffer form (two tiles in flight; fits the 160 KiB at 27 x 27 by 128
// bytes, exchange area single-buffered behind a second barrier) measured the same (242 us) and is not kept: what remains is the
// latency chain of a tile's softmax inside a wave (LDS exchange -> positional lookups -> max / exp / sum -> hi/lo split -> transposed
// reads -> P.x) with two waves per SIMD to hide it.
// ---------------------------------------------------------------------------------------------
// Round 4: the softmax is DE-DUPLICATED.  Until then every one of the four channel-slice waves of a row group redid the whole tile's
// softmax (logit exchange, positional lookups, max / exp / sum, hi / lo split: ~200 VALU instructions per wave and tile, four times
// over) because each needs P as its MFMA operand.  Now a wave owns a QUARTER of the tile's rows: lane (row, token) holds one logit,
// row max / sum are DPP row reductions, and P (bf16 hi | lo, already in A-operand order) and the rescale factors go through 2 KB
// of LDS to all four waves -- a second barrier per tile again, a quarter of the VALU work: 248 -> 233 us alone at C2 (tools/wide_bench.py).
// Same arithmetic per element (same fp32 logits, same exp, same split).  What is left is the tile's latency chain
// (LDS exchange -> lookups -> DPP reductions -> exp -> LDS hand-over -> transposed reads -> MFMAs, ~5k clocks per 16 tokens, both
// waves of a SIMD in the same phase): two tiles per barrier interval would overlap two such chains, and 160 KB of LDS hold four
// 36-KB tiles, not the six that needs.
// BWD = true: the attention backward over the same stream (see global_stream_kernel<NB, true>): the "queries" are the upstream
// gradients dctx_r, the score tile is dP, the quarter-of-the-rows step forms dS = exp(S - M) / L (dP - delta) from the forward's logits
// (fetched one tile ahead by an asm load the end-of-iteration wait covers: the loading waves' counted LDS-DMA and compiler-counted
// loads do not mix), dS goes through the same P hand-over into the dS . x MFMAs, no running max, no rescale.  The positional
// marginals of dS (part_marg) are what the positional part of d q~ needs: with them the [rows, N] dS tensor need not be written.
template <int NB, bool BWD = false>
__global__ __launch_bounds__(512, 1) void global_stream_wide_kernel(StreamParams p) {
    constexpr int NBUF = 3, NRED = 2;
    constexpr int E = NB * 128;
    constexpr int SLICE = E / 4;
    constexpr int KSTEPS = SLICE / 32;
    constexpr int CBLK = SLICE / 16;
    constexpr int TILE_BYTES = NB * 4096;
    constexpr int PIECES = NB * 4;
    constexpr int PPW = PIECES / 4;       // DMA pieces per (loading) wave and tile
    constexpr int RG = kWideRG;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tilebuf = smem;                                                    // [NBUF][TILE_BYTES]
    float* red = reinterpret_cast<float*>(smem + NBUF * TILE_BYTES);        // [NRED][RG][4 waves][16 rows][16 tokens]
    int* tokpos = reinterpret_cast<int*>(red + NRED * RG * 4 * 256);        // [2][16] packed (frame - f_first) << 16 | y << 8 | x
    uint16_t* pbuf = reinterpret_cast<uint16_t*>(tokpos + 32);              // [RG][16 rows][4 k groups][8] P of the current tile, A-operand order (hi x 4 | lo x 4)
    float* alpha_s = reinterpret_cast<float*>(pbuf + RG * 16 * 4 * 8);       // [RG][16] rescale factors of the current tile
    float* postab = alpha_s + RG * 16;                                       // [RG * 16][kWideFrames + H + W]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;       // row group of this wave, channel slice of this wave
    const int r16 = lane & 15, kg = lane >> 4;
    const int part = blockIdx.x, rg = blockIdx.y * RG + grp, nparts = gridDim.x;
    const int tb = (int)(((long)p.ntiles * part) / nparts);
    const int te = (int)(((long)p.ntiles * (part + 1)) / nparts);
    const int S = kWideFrames + p.H + p.W;
    const unsigned f_first = (unsigned)(tb * 16) / (unsigned)p.HW;
    const long row_glob = (long)rg * 16 + r16;
    const unsigned tokpos_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const int*)(tokpos);
    const unsigned tile_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tilebuf));

    bf16x8 ahi[KSTEPS], alo[KSTEPS];
    {
        const long off = row_glob * E + SLICE * wave + 8 * kg;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            ahi[s] = *reinterpret_cast<const bf16x8*>(p.qhi + off + 32 * s);
            alo[s] = *reinterpret_cast<const bf16x8*>(p.qlo + off + 32 * s);
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(ahi[s]), "+v"(alo[s]));   // landed before any DMA is issued
    }
    if (p.pos_a) {
        for (int e = tid; e < RG * 16 * S; e += 512) {
            const int r = e / S, c = e - r * S;
            const long row = (long)(blockIdx.y * RG) * 16 + r;
            int col;
            if (c < kWideFrames) col = p.t0i + (int)f_first + c;
            else if (c < kWideFrames + p.H) col = p.y0i + (c - kWideFrames);
            else col = p.x0i + (c - kWideFrames - p.H);
            postab[e] = col < p.pos_stride ? p.pos_a[row * p.pos_stride + col] : 0.f;
        }
    }

    f32x4 acc[CBLK];
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // online-softmax state of row 4 * wave + (lane >> 4) of this wave's row group (replicated over the 16 token lanes of a DPP row):
    // each of the four waves of a row group owns a QUARTER of the tile's rows
    float m_run = -1.0e30f, l_run = 0.f;
    const int rl = lane >> 4, tok = lane & 15, srow = 4 * wave + rl;
    // backward: softmax state and delta of this lane's row, the forward's logit of (row, token) one tile ahead
    float bw_m = 0.f, bw_linv = 0.f, bw_delta = 0.f, s_nx = 0.f;
    const long srow_g = (long)rg * 16 + srow;
    if constexpr (BWD) {
        if (srow_g < p.rows) {
            bw_m = p.ml[2 * srow_g];
            bw_linv = 1.0f / p.ml[2 * srow_g + 1];
            bw_delta = p.delta[srow_g];
        }
        asm volatile("" : "+v"(bw_m), "+v"(bw_linv), "+v"(bw_delta));      // landed before any DMA is issued
    }
    auto fetch_s = [&](int tile) {                                         // (waited for by the fence at the end of the iteration)
        const float* src = p.s_in + srow_g * p.score_stride + (long)tile * 16 + tok;
        asm volatile("global_load_dword %0, %1, off" : "=v"(s_nx) : "v"(src) : "memory");
    };

    // positional marginals in the kernel (part_marg given): P[16 rows x 16 tokens] . onehot[16 tokens x 16 columns] on the matrix
    // pipe, one MFMA per 16-column block of [frame | grid row | grid column]; the blocks are dealt to the 4 waves of the row group
    // (wave w: blocks w, w + 4; <= 8 blocks: hicom_global_stream_has_marg).  The one-hot B operand is four compares on the token
    // coordinates the lane already holds for the score-side lookups (same token <-> k-slot map as P).
    constexpr int MBW = kWideMargBlocks;
    const bool marg_on = p.part_marg != nullptr;
    const int nby = (p.H + 15) >> 4, nbx = (p.W + 15) >> 4, nmb = marg_on ? 1 + nby + nbx : 0;
    int mb_sh[MBW], mb_base[MBW];
#pragma unroll
    for (int u = 0; u < MBW; ++u) {
        const int b = wave + 4 * u;
        mb_sh[u] = b == 0 ? 16 : (b <= nby ? 8 : 0);
        mb_base[u] = b == 0 ? 0 : (b <= nby ? 16 * (b - 1) : 16 * (b - 1 - nby));       // (wave-uniform: scalar registers)
    }
    f32x4 macc[MBW];
#pragma unroll
    for (int u = 0; u < MBW; ++u) macc[u] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int tile, int buf) {
        if (grp != 0) return;                            // waves 0-3 own the vector-memory requests
        const int r = lane >> 4, cpos = lane & 15;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pi = wave + 4 * i;
            const int blk = pi >> 2, row = 4 * (pi & 3) + r;
            long tok = (long)tile * 16 + row;
            tok = tok < p.N ? tok : p.N - 1;
            const char* src = reinterpret_cast<const char*>(p.x) + tok * (long)(E * 2) + blk * 256 + 16 * (cpos ^ swz(row));
            const unsigned dst = tile_lds + buf * TILE_BYTES + pi * 1024;          // wave-uniform LDS base of the piece
            unsigned keep_m0;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_m0) : "v"(src), "s"(dst) : "memory");
        }
    };
    const int q4 = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 4 * sig(kg) + q4;
    const int rd_row_off = r16 * 256, rd_swz = swz(r16);
    const int tr_row_off = trow * 256 + 8 * (pp & 1), tr_swz = swz(trow);

    // partial scores of this wave's row group over its channel slice, for the tile in `img` (MFMAs only: nothing waits here)
    auto scores_of = [&](const char* img) {
        f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int ch0 = SLICE * wave + 32 * s;
            const int blk = ch0 >> 7, cbase = (ch0 & 127) >> 3;
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(img + blk * 4096 + rd_row_off + 16 * ((cbase + kg) ^ rd_swz));
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[s], b, s4, 0, 0, 0);
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[s], b, s4, 0, 0, 0);
        }
        return s4;
    };
    // publish the partials of `tile` (half `h` of the exchange area) and its token coordinates
    auto publish = [&](const f32x4& s4, int tile, int h) {
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // MFMA -> VALU read (see fused_ring.hip)
        float* rw = red + (h * RG * 4 + grp * 4 + wave) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) rw[(4 * kg + j) * 16 + r16] = s4[j];
        if (p.pos_a && wave8 == 7 && lane < 16) {
            long nn = (long)tile * 16 + lane;
            nn = nn < p.N ? nn : p.N - 1;
            const unsigned un = (unsigned)nn;
            const unsigned t = un / (unsigned)p.HW, rem = un - t * (unsigned)p.HW;
            const unsigned y = rem / (unsigned)p.W, xx = rem - y * (unsigned)p.W;
            unsigned tf = t - f_first;
            tf = tf < (unsigned)kWideFrames ? tf : kWideFrames - 1;             // (only masked tail tokens clamp)
            tokpos[16 * h + lane] = (int)((tf << 16) | (y << 8) | xx);
        }
    };
    // the loading waves' pieces of the youngest staged tile have landed (nothing may stay in flight across the barrier: the next
    // iteration's scores read that tile), then the workgroup barrier
#ifdef HICOM_WTRACE
    unsigned long long wt_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, wt_last = 0;
#endif
    auto fence = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        HICOM_WT(8);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        HICOM_WT(9);
    };

    if constexpr (BWD) {
        if (tb < te) fetch_s(tb);
    }
    if (tb < te) stage(tb, 0);
    if (tb + 1 < te) stage(tb + 1, 1);
    __syncthreads();                                   // positional tables
    fence();                                           // tiles tb, tb + 1 landed
    if (tb < te) {
        const f32x4 s0 = scores_of(tilebuf);
        publish(s0, tb, 0);
    }
    fence();                                           // partials of tile tb published

    for (int tile = tb; tile < te; ++tile) {
        const int k = tile - tb, cur = k % NBUF, h = k & 1;
        const bool more = tile + 1 < te;
        float s_cur = 0.f;
        if constexpr (BWD) {
            s_cur = s_nx;                                                 // (landed: the fence of the previous iteration waited for it)
            asm volatile("" : "+v"(s_cur));
            if (more) fetch_s(tile + 1);
        }
#ifdef HICOM_WTRACE
        if (k == 0) { wt_last = __builtin_readcyclecounter(); for (int i_ = 0; i_ < 12; ++i_) wt_sum[i_] = 0; }
#endif
        if (tile + NBUF - 1 < te) stage(tile + NBUF - 1, (k + NBUF - 1) % NBUF);     // into the buffer of tile - 1 (retired by the barrier just passed)
        HICOM_WT(1);
        const char* img = tilebuf + cur * TILE_BYTES;
        // ---- scores of tile + 1: issued now, needed at the end of the iteration ----
        f32x4 sn = f32x4{0.f, 0.f, 0.f, 0.f};
        if (more) sn = scores_of(tilebuf + ((k + 1) % NBUF) * TILE_BYTES);
        __builtin_amdgcn_sched_barrier(0);
        HICOM_WT(2);

        // ---- softmax of THIS tile, a quarter of the rows per wave: lane (row srow, token tok) holds ONE logit ----
        {
            const float* rb = red + (h * RG * 4 + grp * 4) * 256 + srow * 16 + tok;
            float lg = (rb[0] + rb[256]) + (rb[512] + rb[768]);
            if (p.pos_a) {
                const int tp = tokpos[16 * h + tok];
                const float* pr_ = postab + (grp * 16 + srow) * S;
                lg += pr_[tp >> 16] + pr_[kWideFrames + ((tp >> 8) & 255)] + pr_[kWideFrames + p.H + (tp & 255)];
            }
            const long n = (long)tile * 16 + tok;
            const bool ok = n < p.N;
            float pr, alpha = 1.0f;
            if constexpr (BWD) {
                // dS = softmax weight x (dP - delta); padded rows (state 0, 0, 0) and tail tokens contribute nothing
                pr = (ok && srow_g < p.rows) ? expf(s_cur - bw_m) * bw_linv * (lg - bw_delta) : 0.f;
                if (p.scores) p.scores[srow_g * p.score_stride + n] = pr;
            } else {
                if (p.scores) p.scores[srow_g * p.score_stride + n] = lg;
                const float m_new = fmaxf(m_run, row16_max(ok ? lg : -1.0e30f));
                alpha = fast_exp(m_run - m_new);
                pr = ok ? fast_exp(lg - m_new) : 0.f;
                l_run = l_run * alpha + row16_sum(pr);
                m_run = m_new;
            }
            uint16_t hh, ll;
            split_bf16(pr, hh, ll);
            uint16_t* pb = pbuf + ((grp * 16 + srow) * 4 + sig(tok >> 2)) * 8 + (tok & 3);   // k slot 8 kg + j <-> token 4 sig(kg) + j
            pb[0] = hh;
            pb[4] = ll;
            if (!BWD && tok == 0) alpha_s[grp * 16 + srow] = alpha;
        }
        HICOM_WT(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HICOM_WT(4);
        __builtin_amdgcn_s_barrier();                    // P and the rescale factors of this tile are complete
        asm volatile("" ::: "memory");
        HICOM_WT(5);

        // ---- this wave's channel slice: rescale, marginals, P.x ----
        const bf16x8 pw = *reinterpret_cast<const bf16x8*>(pbuf + ((grp * 16 + r16) * 4 + kg) * 8);
        f32x4 al = f32x4{1.f, 1.f, 1.f, 1.f};
        if constexpr (!BWD) al = *reinterpret_cast<const f32x4*>(alpha_s + grp * 16 + 4 * kg);
        int tp[4] = {0, 0, 0, 0};
        if (marg_on) {
            int4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(tokpos_lds + 64 * h + 16 * sig(kg)) : "memory");
            tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
        }
        if (!BWD && __any(al[0] != 1.0f || al[1] != 1.0f || al[2] != 1.0f || al[3] != 1.0f)) {
#pragma unroll
            for (int cb = 0; cb < CBLK; ++cb) {
                acc[cb][0] *= al[0]; acc[cb][1] *= al[1]; acc[cb][2] *= al[2]; acc[cb][3] *= al[3];
            }
#pragma unroll
            for (int u = 0; u < MBW; ++u) {
                macc[u][0] *= al[0]; macc[u][1] *= al[1]; macc[u][2] *= al[2]; macc[u][3] *= al[3];
            }
        }
#pragma unroll
        for (int u = 0; u < MBW; ++u) {
            if (wave + 4 * u < nmb) {
                bf16x8 oh;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const short one = (((tp[j] >> mb_sh[u]) & 255) - mb_base[u] == r16) ? (short)0x3F80 : (short)0;
                    oh[j] = one;
                    oh[4 + j] = one;
                }
                macc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw, oh, macc[u], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned img_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(img);
        constexpr int PGW = (CBLK % 6 == 0) ? 6 : CBLK;      // transposed fragments in flight together
#pragma unroll
        for (int c0 = 0; c0 < CBLK; c0 += PGW) {
            bf16x4 bvs[PGW];
#pragma unroll
            for (int u = 0; u < PGW; ++u) {
                const int ch0 = SLICE * wave + 16 * (c0 + u);
                const int blk = ch0 >> 7, c2 = (ch0 & 127) >> 3;
                const unsigned a = img_lds + blk * 4096 + tr_row_off + 16 * ((c2 + (pp >> 1)) ^ tr_swz);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bvs[u]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PGW; ++u) {
                const bf16x8 b2 = bf16x8{bvs[u][0], bvs[u][1], bvs[u][2], bvs[u][3], bvs[u][0], bvs[u][1], bvs[u][2], bvs[u][3]};
                acc[c0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw, b2, acc[c0 + u], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        HICOM_WT(6);
        // ---- hand the next tile's partials over ----
        if (more) publish(sn, tile + 1, h ^ 1);
        HICOM_WT(7);
        fence();                                         // (tile + 2, staged in this iteration, has landed: the next scores read it)
    }
#ifdef HICOM_WTRACE
    if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 512) {
        wt_sum[0] = (unsigned long long)(te - tb);
        for (int i_ = 0; i_ < 12; ++i_) g_wide_trace[(blockIdx.x * 8 + wave8) * 12 + i_] = wt_sum[i_];
    }
#endif

    const long prow = (long)part * p.rows_pad + rg * 16;
    if (!BWD && tok == 0 && rg * 16 + srow < p.rows) {
        p.part_m[prow + srow] = m_run;
        p.part_l[prow + srow] = l_run;
    }
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) {
        float* o = p.part_acc + (prow + 4 * kg) * E + SLICE * wave + 16 * cb + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rg * 16 + 4 * kg + j < p.rows) o[(long)j * E] = acc[cb][j];
    }
#pragma unroll
    for (int u = 0; u < MBW; ++u) {
        if (wave + 4 * u < nmb) {
            float* o = p.part_marg + (prow + 4 * kg) * p.marg_stride + 16 * (wave + 4 * u) + r16;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (rg * 16 + 4 * kg + j < p.rows) o[(long)j * p.marg_stride] = macc[u][j];
        }
    }
}

#ifdef HICOM_WTRACE
}  // namespace hicom
extern "C" int hicom_debug_wide_trace(void* dst, int64_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(hicom::g_wide_trace), (size_t)bytes) == hipSuccess ? HICOM_OK : HICOM_ELAUNCH;
}
namespace hicom {
#endif

static int g_num_cus = 0;
static int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_global_stream_nparts(int64_t N, int32_t rows_pad) {
    if (N <= 0 || rows_pad <= 0) return HICOM_EINVAL;
    const long ntiles = (N + 15) / 16;
    const int groups = rows_pad / 16;
    long want = 2L * num_cus() / (groups > 0 ? groups : 1);   // 2 resident workgroups per CU
    if (rows_pad % (16 * kWideRG) == 0 && rows_pad > 16) want = (long)num_cus() / (groups / kWideRG);   // wide form: one per CU
    if (want < 1) want = 1;
    // keep >= 4 tiles per chunk so the per-chunk partial write stays a small fraction of the stream
    long cap = ntiles / 4;
    if (cap < 1) cap = 1;
    return (int)(want < cap ? want : cap);
}

extern "C" int hicom_global_stream_marg_width(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return HICOM_EINVAL;
    return 16 * (1 + (H + 15) / 16 + (W + 15) / 16);
}

static bool wide_ok(int64_t N, int32_t E, int32_t rows_pad, int32_t H, int32_t W, int32_t nparts) {
    const bool pos = H > 0 && W > 0;
    const long ntiles = (N + 15) / 16, HW = pos ? (long)H * W : 1;
    const long tiles_per_part = (ntiles + nparts - 1) / nparts + 1;
    const bool span_ok = !pos || (tiles_per_part * 16 + HW - 1) / HW + 1 <= kWideFrames;
    return E == 1152 && rows_pad > 16 && rows_pad % (16 * kWideRG) == 0 && (!pos || (H <= 64 && W <= 64)) && span_ok;
}

// 1 when hicom_global_stream_marg_fwd applies to this shape: the many-row form of the kernel (the only one with in-kernel
// positional marginals) and a grid whose [frames | rows | columns] fit its 8 marginal blocks of 16 columns
extern "C" int hicom_global_stream_has_marg(int64_t N, int32_t E, int32_t rows_pad, int32_t H, int32_t W, int32_t nparts) {
    if (N <= 0 || nparts <= 0 || rows_pad <= 0 || H <= 0 || W <= 0) return HICOM_EINVAL;
    const char* force_narrow = getenv("HICOM_GLOBAL_NARROW");      // dev / test switch: always take the one-row-group kernel
    if (force_narrow && force_narrow[0] == '1') return 0;
    const char* no_marg = getenv("HICOM_GLOBAL_NO_MARG");          // dev / A-B switch: logit tensor + marginal pass (the round-3 path)
    if (no_marg && no_marg[0] == '1') return 0;
    // (the merge behind it keeps its weight record -- 2 + nparts + T + H + W floats per row -- in the head of the row's scratch region of
    // T (H + W + 2) floats, hicom_global_merge_marg_fwd: a single image, T = 1, has no room for it and takes the logit-tensor form)
    if (N % ((long)H * W) != 0) return 0;
    const long T = N / ((long)H * W);
    if (2 + (long)nparts + T + H + W > T * (H + W + 2)) return 0;
    return (wide_ok(N, E, rows_pad, H, W, nparts) && 1 + (H + 15) / 16 + (W + 15) / 16 <= 4 * kWideMargBlocks) ? 1 : 0;
}

static int global_stream_launch(const void* x, int64_t N, int32_t E,
                                       const void* qt_hi, const void* qt_lo, int32_t rows, int32_t rows_pad,
                                       const float* pos_a, int32_t pos_stride,
                                       int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                       float* scores, int64_t score_stride,
                                       float* part_m, float* part_l, float* part_acc, int32_t nparts,
                                       const float* inv_norm, const float* row_const, float* part_marg, void* stream) {
    HICOM_REQUIRE(x && qt_hi && qt_lo && (scores || part_marg) && part_m && part_l && part_acc, HICOM_EINVAL, "global_stream: NULL pointer");
    HICOM_REQUIRE(E == 1152 || E == 768, HICOM_EUNSUP, "global_stream: E=%d (only 1152 / 768)", E);
    HICOM_REQUIRE(N > 0 && N < (1L << 31), HICOM_EINVAL, "global_stream: N out of range");
    HICOM_REQUIRE(rows_pad > 0 && rows_pad % 16 == 0 && rows > 0 && rows <= rows_pad, HICOM_EINVAL,
                  "global_stream: rows_pad must be a multiple of 16 and 0 < rows <= rows_pad");
    HICOM_REQUIRE(nparts > 0, HICOM_EINVAL, "global_stream: nparts");
    HICOM_REQUIRE(!scores || (score_stride >= ((N + 15) / 16) * 16 && score_stride % 4 == 0), HICOM_EINVAL,
                  "global_stream: score_stride must be >= roundup(N,16)");
    HICOM_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)qt_hi % 16 == 0) && ((uintptr_t)qt_lo % 16 == 0) &&
                      ((uintptr_t)scores % 16 == 0),
                  HICOM_EINVAL, "global_stream: pointers must be 16-byte aligned");
    if (pos_a) HICOM_REQUIRE(H > 0 && W > 0 && pos_stride > 0, HICOM_EINVAL, "global_stream: pos geometry");
    StreamParams p;
    p.x = (const uint16_t*)x; p.N = N; p.qhi = (const uint16_t*)qt_hi; p.qlo = (const uint16_t*)qt_lo;
    p.pos_a = pos_a; p.pos_stride = pos_stride; p.H = H; p.W = W; p.HW = H * W;
    p.t0i = t_index0; p.y0i = y_index0; p.x0i = x_index0;
    p.scores = scores; p.score_stride = score_stride;
    p.part_m = part_m; p.part_l = part_l; p.part_acc = part_acc; p.rows_pad = rows_pad; p.rows = rows;
    p.ntiles = (int)((N + 15) / 16);
    p.s_in = nullptr; p.ml = nullptr; p.delta = nullptr;
    p.inv_norm = inv_norm; p.row_const = row_const;
    p.part_marg = pos_a ? part_marg : nullptr;
    p.marg_stride = pos_a ? hicom_global_stream_marg_width(H, W) : 0;
    dim3 grid((unsigned)nparts, (unsigned)(rows_pad / 16));
    hipStream_t s = (hipStream_t)stream;
    // many rows: the wide form (two row groups per workgroup, three-deep ring) when its limits hold
    const char* force_narrow = getenv("HICOM_GLOBAL_NARROW");      // dev / test switch: always take the one-row-group kernel
    const bool wide = !inv_norm && !(force_narrow && force_narrow[0] == '1') && wide_ok(N, E, rows_pad, pos_a ? H : 0, pos_a ? W : 0, nparts);
    HICOM_REQUIRE(!part_marg || (wide && hicom_global_stream_has_marg(N, E, rows_pad, H, W, nparts) == 1), HICOM_EUNSUP,
                  "global_stream: no in-kernel marginals for this shape (hicom_global_stream_has_marg)");
    HICOM_REQUIRE(scores || part_marg, HICOM_EINVAL, "global_stream: scores is NULL");
    if (wide) {
        const int S = kWideFrames + (pos_a ? H + W : 0);
        const size_t smem = (size_t)3 * 9 * 4096 + 2 * (size_t)kWideRG * 4 * 1024 + 128 + (size_t)kWideRG * (1024 + 64) + (size_t)kWideRG * 16 * S * 4;
        static bool wide_attr = false;
        if (!wide_attr) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(global_stream_wide_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            wide_attr = true;
        }
        if (!pos_a) { p.H = 1; p.W = 1; p.HW = 1; }
        hipLaunchKernelGGL(global_stream_wide_kernel<9>, dim3((unsigned)nparts, (unsigned)(rows_pad / (16 * kWideRG))), dim3(512), smem, s, p);
        return hicom_host::check_launch("global_stream");
    }
    if (E == 1152) {
        constexpr int smem = 2 * 9 * 4096 + 4096;
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(global_stream_kernel<9>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            attr_set = true;
        }
        hipLaunchKernelGGL(global_stream_kernel<9>, grid, dim3(256), smem, s, p);
    } else {
        constexpr int smem = 2 * 6 * 4096 + 4096;
        hipLaunchKernelGGL(global_stream_kernel<6>, grid, dim3(256), smem, s, p);
    }
    return hicom_host::check_launch("global_stream");
}

extern "C" int hicom_global_stream_fwd(const void* x, int64_t N, int32_t E,
                                       const void* qt_hi, const void* qt_lo, int32_t rows, int32_t rows_pad,
                                       const float* pos_a, int32_t pos_stride,
                                       int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                       float* scores, int64_t score_stride,
                                       float* part_m, float* part_l, float* part_acc, int32_t nparts,
                                       void* stream) {
    return global_stream_launch(x, N, E, qt_hi, qt_lo, rows, rows_pad, pos_a, pos_stride, H, W, t_index0, y_index0, x_index0, scores,
                                score_stride, part_m, part_l, part_acc, nparts, nullptr, nullptr, nullptr, stream);
}

// Same stream, positional marginals of the softmax weights accumulated IN the kernel (part_marg: f32 [nparts][rows_pad]
// [hicom_global_stream_marg_width(H, W)], merged by hicom_global_merge_marg_fwd): `scores` may be NULL -- the [rows_pad, N] logit
// tensor (54 MB at 288 rows x 64 frames) is then neither written nor read back.  Wide form only (hicom_global_stream_has_marg).
extern "C" int hicom_global_stream_marg_fwd(const void* x, int64_t N, int32_t E,
                                            const void* qt_hi, const void* qt_lo, int32_t rows, int32_t rows_pad,
                                            const float* pos_a, int32_t pos_stride,
                                            int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                            float* scores, int64_t score_stride,
                                            float* part_m, float* part_l, float* part_acc, float* part_marg, int32_t nparts,
                                            void* stream) {
    HICOM_REQUIRE(pos_a && part_marg, HICOM_EINVAL, "global_stream_marg: pos_a and part_marg are required");
    return global_stream_launch(x, N, E, qt_hi, qt_lo, rows, rows_pad, pos_a, pos_stride, H, W, t_index0, y_index0, x_index0, scores,
                                score_stride, part_m, part_l, part_acc, nparts, nullptr, nullptr, part_marg, stream);
}

extern "C" int hicom_global_stream_clip_fwd(const void* x, int64_t N, int32_t E,
                                            const void* qt_hi, const void* qt_lo, int32_t rows, int32_t rows_pad,
                                            const float* pos_a, int32_t pos_stride,
                                            int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                            const float* inv_norm, const float* row_const,
                                            float* scores, int64_t score_stride,
                                            float* part_m, float* part_l, float* part_acc, int32_t nparts, void* stream) {
    HICOM_REQUIRE(inv_norm && row_const, HICOM_EINVAL, "global_stream_clip: NULL pointer");
    return global_stream_launch(x, N, E, qt_hi, qt_lo, rows, rows_pad, pos_a, pos_stride, H, W, t_index0, y_index0, x_index0, scores,
                                score_stride, part_m, part_l, part_acc, nparts, inv_norm, row_const, nullptr, stream);
}

// ---- attention backward over the stream (training path) --------------------------------------------
// dctx_hi / dctx_lo : bf16 [rows_pad, E] hi / lo planes of the upstream gradients of the per-head contexts (rows >= rows zero)
// pos_b             : f32 [rows_pad, pos_stride] = dctx . PE^T (score-side table of dP), or NULL
// s_in              : the logits the forward pass (hicom_global_stream_fwd) wrote; ml [rows][2]; delta [rows]
// ds_out            : f32 [rows_pad, score_stride] dS; part_acc : f32 [nparts, rows_pad, E] partial sums of dS . x
extern "C" int hicom_global_stream_bwd(const void* x, int64_t N, int32_t E, const void* dctx_hi, const void* dctx_lo,
                                       int32_t rows, int32_t rows_pad, const float* pos_b, int32_t pos_stride,
                                       int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                       const float* s_in, int64_t score_stride, const float* ml, const float* delta,
                                       float* ds_out, float* part_acc, float* part_marg, int32_t nparts, void* stream) {
    HICOM_REQUIRE(x && dctx_hi && dctx_lo && s_in && ml && delta && (ds_out || part_marg || !pos_b) && part_acc, HICOM_EINVAL, "global_stream_bwd: NULL pointer");
    HICOM_REQUIRE(E == 1152 || E == 768, HICOM_EUNSUP, "global_stream_bwd: E=%d (only 1152 / 768)", E);
    HICOM_REQUIRE(N > 0 && N < (1L << 31) && rows_pad > 0 && rows_pad % 16 == 0 && rows > 0 && rows <= rows_pad && nparts > 0,
                  HICOM_EINVAL, "global_stream_bwd: bad shape");
    HICOM_REQUIRE(score_stride >= ((N + 15) / 16) * 16 && score_stride % 4 == 0, HICOM_EINVAL, "global_stream_bwd: score_stride");
    HICOM_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)dctx_hi % 16 == 0) && ((uintptr_t)dctx_lo % 16 == 0) &&
                      ((uintptr_t)s_in % 16 == 0) && ((uintptr_t)ds_out % 16 == 0), HICOM_EINVAL, "global_stream_bwd: alignment");
    HICOM_REQUIRE(rows_pad % 16 == 0, HICOM_EINVAL, "global_stream_bwd: rows_pad");
    if (pos_b) HICOM_REQUIRE(H > 0 && W > 0 && pos_stride > 0, HICOM_EINVAL, "global_stream_bwd: pos geometry");
    StreamParams p;
    p.x = (const uint16_t*)x; p.N = N; p.qhi = (const uint16_t*)dctx_hi; p.qlo = (const uint16_t*)dctx_lo;
    p.pos_a = pos_b; p.pos_stride = pos_stride; p.H = H; p.W = W; p.HW = H * W;
    p.t0i = t_index0; p.y0i = y_index0; p.x0i = x_index0;
    p.scores = ds_out; p.score_stride = score_stride;
    p.part_m = nullptr; p.part_l = nullptr; p.part_acc = part_acc; p.rows_pad = rows_pad; p.rows = rows;
    p.ntiles = (int)((N + 15) / 16);
    p.s_in = s_in; p.ml = ml; p.delta = delta; p.inv_norm = nullptr; p.row_const = nullptr; p.part_marg = nullptr; p.marg_stride = 0;
    dim3 grid((unsigned)nparts, (unsigned)(rows_pad / 16));
    hipStream_t s = (hipStream_t)stream;
    // many rows: the wide form (two row groups per workgroup, half the passes over the tokens; with part_marg the positional marginals
    // of dS come out of the kernel and ds_out may be NULL)
    const char* force_narrow = getenv("HICOM_GLOBAL_NARROW");
    const bool wide = !(force_narrow && force_narrow[0] == '1') && wide_ok(N, E, rows_pad, pos_b ? H : 0, pos_b ? W : 0, nparts);
    HICOM_REQUIRE(!part_marg || (wide && pos_b && hicom_global_stream_has_marg(N, E, rows_pad, H, W, nparts) == 1), HICOM_EUNSUP,
                  "global_stream_bwd: no in-kernel marginals for this shape (hicom_global_stream_has_marg)");
    HICOM_REQUIRE(wide || ds_out, HICOM_EINVAL, "global_stream_bwd: ds_out is NULL");
    if (wide) {
        const int S = kWideFrames + (pos_b ? H + W : 0);
        const size_t smem = (size_t)3 * 9 * 4096 + 2 * (size_t)kWideRG * 4 * 1024 + 128 + (size_t)kWideRG * (1024 + 64) + (size_t)kWideRG * 16 * S * 4;
        static bool wide_attr = false;
        if (!wide_attr) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(global_stream_wide_kernel<9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            wide_attr = true;
        }
        p.part_marg = part_marg;
        p.marg_stride = part_marg ? hicom_global_stream_marg_width(H, W) : 0;
        if (!pos_b) { p.H = 1; p.W = 1; p.HW = 1; }
        hipLaunchKernelGGL((global_stream_wide_kernel<9, true>), dim3((unsigned)nparts, (unsigned)(rows_pad / (16 * kWideRG))), dim3(512), smem, s, p);
        return hicom_host::check_launch("global_stream_bwd");
    }
    if (E == 1152) {
        constexpr int smem = 2 * 9 * 4096 + 4096;
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(global_stream_kernel<9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            attr_set = true;
        }
        hipLaunchKernelGGL((global_stream_kernel<9, true>), grid, dim3(256), smem, s, p);
    } else {
        constexpr int smem = 2 * 6 * 4096 + 4096;
        hipLaunchKernelGGL((global_stream_kernel<6, true>), grid, dim3(256), smem, s, p);
    }
    return hicom_host::check_launch("global_stream_bwd");
}
