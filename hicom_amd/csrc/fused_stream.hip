// Fused hybrid-level compressor stream: LOCAL windowed attention + GLOBAL multi-head attention in
// ONE pass over the visual tokens -- frames_embed and frames_feature are each read from HBM
// exactly once ("local + global share the value stream").
//
// Replaces, for the release recipe (use_guide = direct, exact window partition), both
//   LocalCompressor.forward  windows/bmm/softmax/bmm   (reference projector.py:544-558) and
//   MultiheadAttention.forward QK^T/softmax/PV          (reference projector.py:193-215).
//
// One 16-row MFMA operand carries BOTH problems:
//   rows 0 .. R-1   : folded global queries qt_h (bf16 hi + lo), scored against frames_feature
//   rows R .. 15    : the local query (guide), scored against frames_embed; row R + (w mod NLOC)
//                     belongs to window w, other windows' tokens are masked out of that row
// so the P.x product (v_mfma_f32_16x16x16_bf16 against the frames_feature tile) accumulates the 9
// global head contexts AND the open windows' local contexts in the same 72 accumulator VGPRs.
//
// Token order: a workgroup walks a contiguous range of windows in window-major order (the 36
// tokens of a window, then the next window), 16 tokens per tile.  A window spans <= 4 tiles and
// at most 2 windows are open per tile, so NLOC = 16 - R >= 2 local rows suffice; when a window's
// last token has been accumulated its row is normalised, written to ctx_local and recycled.
//
// Per tile: frames_feature rows -> LDS by LDS-DMA (same swizzled image as global_stream.hip:
// row reads for the score B operand, transposed reads for the P.x B operand); frames_embed goes
// straight from HBM into B-fragment registers (it is only needed once, by the local score).
// Positional logit terms come from small per-workgroup LDS tables (a_t for the <= 16 frames a
// workgroup touches, a_y, a_x), so the tile loop contains no ordinary global load whose in-order
// vmcnt would drain the LDS-DMA prefetch.
#include <stdlib.h>

#include "common.hpp"

namespace hicom {

// Dev-only phase timeline (tools/fused_trace.py builds a second library with -DHICOM_TRACE): lane 0 of
// every workgroup stamps s_memtime at the phase boundaries.  Compiled out of the product.
#ifdef HICOM_TRACE
__device__ unsigned long long g_fused_trace[1024 * 128];
constexpr int kTraceStores = 1;   // trace stores between the pos requests and their counted wait
#define HICOM_TR() do { if (tid == 0 && tr_n < 128) g_fused_trace[blockIdx.x * 128 + tr_n] = __builtin_readcyclecounter(); ++tr_n; } while (0)
#else
constexpr int kTraceStores = 0;
#define HICOM_TR() do {} while (0)
#endif

constexpr int kMaxWinPerWg = 16;
constexpr int kMaxFramesPerWg = 8;
constexpr int kPosPerThread = 3;     // score-side pos-emb table entries fetched per thread: rows * (8 + H + W) <= 768
constexpr int kPStride = 32;        // halfwords between rows of the softmax-weight planes (they share red[0]'s rows)
constexpr int kMargW = 12;         // per (row, window): kt + 2 * ks <= 11 marginal bins + the reference max in slot 11

struct FusedParams {
    const uint16_t* ff;
    const uint16_t* fe;
    int T, H, W;
    int kt, ks, nwy, nwx, NW, WSZ;
    const uint16_t* qhi;   // [16][E]  rows < R: qt hi ; rows >= R: local query (exact bf16)
    const uint16_t* qlo;   // [16][E]  rows < R: qt lo ; rows >= R: zero
    int R;
    float l_scale, l_bias;
    const float* pos_a;    // [16][pos_stride] or NULL
    int pos_stride, t0i, y0i, x0i;
    float* part_m;
    float* part_l;
    float* part_acc;       // [nparts][16][E], rows < R
    float* part_marg;      // [nparts][R][wpw][kMargW]: per-window t / y / x marginals of the global softmax weights
                           // (relative to part_m), or NULL
    float* ctx_local;      // [NW][E] fp32 window contexts (may be NULL)
    uint16_t* ctx_hi;      // [NW][E] the same as bf16 hi / lo planes for hicom_planes_gemm_fwd (may be NULL)
    uint16_t* ctx_lo;
    int wpw;               // windows per workgroup
               // developer ablation mask (HICOM_FUSED_DBG): 1 no P.x, 2 no score MFMAs, 8 no fe loads, 16 no LDS-DMA after tile 0, 64 no partial write-out, 128 no score stores, 256 one tile only
};

__device__ __forceinline__ int fswz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int fsig(int g) { return ((g & 1) << 1) | (g >> 1); }

template <int NB>
__global__ __launch_bounds__(256, 2) void fused_stream_kernel(FusedParams p) {
    constexpr int E = NB * 128;
    constexpr int SLICE = E / 4;
    constexpr int KSTEPS = SLICE / 32;
    constexpr int CBLK = SLICE / 16;
    constexpr int TILE_BYTES = NB * 4096;
    constexpr int PIECES = NB * 4;
    static_assert(KSTEPS % 3 == 0, "three score accumulators");
    static_assert(PIECES % 4 == 0, "pieces are dealt round-robin to the 4 waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tilebuf = smem;                                              // [2][TILE_BYTES]
    float* red = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);     // [4 waves][16][16] logit partials (channel slices)
    // The softmax weights (bf16 hi / lo, [16 rows][16 slots] each) live INSIDE red[0]: row r's 64 bytes hold
    // p_hi[r] | p_lo[r].  Only the wave that owns row r reads red[*][r] (after [B]) and it writes P[r] after
    // those reads; red is rewritten after the next [A], when every wave is done reading P.
    uint16_t* p_hi = reinterpret_cast<uint16_t*>(red);                 // row stride kPStride halfwords
    uint16_t* p_lo = p_hi + 16;
    float* mbin = red + 4 * 256;                                       // [2 parities][R][kMargW] marginals of the windows in flight
    float* alpha_s = mbin + (p.part_marg ? 2 * p.R * kMargW : 0);             // [16] rescale factor of each row (this tile)
    float* lrun_s = alpha_s + 16;                                      // [16] running normaliser of each row
    int* win_off = reinterpret_cast<int*>(lrun_s + 16);                // [64] token offset of in-window index
    int* win_txy = win_off + 64;                                       // [64] packed (t2 << 16 | h2 << 8 | w2)
    int* worg = win_txy + 64;                                          // [kMaxWinPerWg] window origin token
    int* wtxy = worg + kMaxWinPerWg;                                   // [kMaxWinPerWg] packed (t0 << 16 | y0 << 8 | x0) window base coords
    float* a_pos = reinterpret_cast<float*>(wtxy + kMaxWinPerWg);      // [R][kMaxFramesPerWg | H | W] score-side pos-emb per row

    const int tid = threadIdx.x, lane = tid & 63;
    int tr_n = 0; (void)tr_n;
    HICOM_TR();   // 0: start
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const int part = blockIdx.x;
    const int wb = part * p.wpw;
    const int we = min(p.NW, wb + p.wpw);
    const int nwin = we - wb;
    const int total = nwin * p.WSZ;                 // tokens of this workgroup's stream
    const int ntile = (total + 15) >> 4;
    const int R = p.R, NLOC = 16 - R;
    const int HW = p.H * p.W, ks2 = p.ks * p.ks;

    // ---- request order of the prologue: (1) the score-side pos-emb table entries (three per thread,
    // L2-resident, tiny), (2) tile 0 of frames_feature by LDS-DMA, (3) its frames_embed fragments, (4) the
    // A operand.  The counted wait below then releases the table build while (2)-(4) are still in
    // flight, instead of the tables waiting behind the first HBM round trip (vmcnt returns in order).
    const int t1_first = wb / (p.nwy * p.nwx);
    float posv[kPosPerThread];
    bool pos_ok[kPosPerThread];
    {
        // branch-free (selects + unconditional loads from a valid address), so the waits stay counted
        const int S = kMaxFramesPerWg + p.H + p.W;                    // entries per row: frames | rows | columns
        const float inv_s = 1.0f / (float)S;
        const float* pa = p.pos_a ? p.pos_a : reinterpret_cast<const float*>(p.qhi);
#pragma unroll
        for (int u = 0; u < kPosPerThread; ++u) {
            const int e = tid + 256 * u;
            const int r = (int)(((float)e + 0.5f) * inv_s), c = e - r * S;
            const int t = t1_first * p.kt + c;
            const int col = c < kMaxFramesPerWg ? p.t0i + t : (c < kMaxFramesPerWg + p.H ? p.y0i + (c - kMaxFramesPerWg) : p.x0i + (c - kMaxFramesPerWg - p.H));
            const bool ok = p.pos_a != nullptr && r < R && (c >= kMaxFramesPerWg || t < p.T);
            // inline asm: hipcc would guard a tracked load's first use with s_waitcnt vmcnt(0) here (the
            // code in between has branches), i.e. behind the whole first HBM round trip
            const float* src = pa + (ok ? (long)r * p.pos_stride + col : 0);
            asm volatile("global_load_dword %0, %1, off" : "=v"(posv[u]) : "v"(src) : "memory");
            pos_ok[u] = ok;
        }
    }
    // tile 0 is requested BEFORE the LDS tables exist (token index by plain arithmetic)
    const unsigned wsz_magic = (65536u + p.WSZ - 1) / p.WSZ;
    auto token_direct = [&](int s) -> long {
        s = s < total ? s : total - 1;
        int wr = (int)(((unsigned)s * wsz_magic) >> 16);
        int i = s - wr * p.WSZ;
        if (i < 0) { i += p.WSZ; wr -= 1; }
        const int w = wb + wr, per_t = p.nwy * p.nwx;
        const int t1 = w / per_t, r = w - t1 * per_t, h1 = r / p.nwx, w1 = r - h1 * p.nwx;
        const int t2 = i / ks2, ri = i - t2 * ks2, h2 = ri / p.ks, w2 = ri - h2 * p.ks;
        return ((long)(t1 * p.kt + t2) * p.H + (h1 * p.ks + h2)) * p.W + (w1 * p.ks + w2);
    };
    bf16x8 bfe[KSTEPS];
    {
        const int row = 4 * wave + (lane >> 4), cpos = lane & 15;
        const char* src = reinterpret_cast<const char*>(p.ff) + token_direct(row) * (long)(E * 2) + 16 * (cpos ^ fswz(row));
#pragma unroll
        for (int i = 0; i < PIECES / 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                             (__attribute__((address_space(3))) void*)(tilebuf + (wave + 4 * i) * 1024), 16, 0, 0);
        const uint16_t* fsrc = p.fe + token_direct(r16) * E + SLICE * wave + 8 * kg;
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) bfe[k] = *reinterpret_cast<const bf16x8*>(fsrc + 32 * k);
    }
    // ---- A operand (hi / lo) of this wave's channel slice ----------------------------------------
    bf16x8 ahi[KSTEPS], alo[KSTEPS];
    {
        const long off = (long)r16 * E + SLICE * wave + 8 * kg;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            ahi[s] = *reinterpret_cast<const bf16x8*>(p.qhi + off + 32 * s);
            alo[s] = *reinterpret_cast<const bf16x8*>(p.qlo + off + 32 * s);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    HICOM_TR();   // prologue: everything requested
    // ---- per-workgroup tables -----------------------------------------------------------------
    if (p.part_marg)
        for (int i = tid; i < 2 * R * kMargW; i += 256) mbin[i] = 0.f;
    if (tid < p.WSZ) {
        const int t2 = tid / ks2, r = tid - t2 * ks2, h2 = r / p.ks, w2 = r - h2 * p.ks;
        win_off[tid] = (t2 * p.H + h2) * p.W + w2;
        win_txy[tid] = (t2 << 16) | (h2 << 8) | w2;
    }
    if (tid < nwin) {
        const int w = wb + tid;
        const int t1 = w / (p.nwy * p.nwx), r = w - t1 * (p.nwy * p.nwx), h1 = r / p.nwx, w1 = r - h1 * p.nwx;
        worg[tid] = (t1 * p.kt * p.H + h1 * p.ks) * p.W + w1 * p.ks;
        wtxy[tid] = (((t1 - t1_first) * p.kt) << 16) | ((h1 * p.ks) << 8) | (w1 * p.ks);
    }
    {
        // the pos entries were the first requests of this wave (in-order return): wait until only the
        // later ones -- tile 0, its frames_embed fragments, the A operand -- are still in flight
        static_assert(kPosPerThread == 3, "operands of the counted wait");
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(posv[0]), "+v"(posv[1]), "+v"(posv[2]) : "n"(PIECES / 4 + 3 * KSTEPS + kTraceStores) : "memory");
        const int n_all = R * (kMaxFramesPerWg + p.H + p.W);
#pragma unroll
        for (int u = 0; u < kPosPerThread; ++u) {
            const int e = tid + 256 * u;
            if (e < n_all) a_pos[e] = pos_ok[u] ? posv[u] : 0.f;
        }
    }

    f32x4 acc[CBLK];
#pragma unroll
    for (int cb = 0; cb < CBLK; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;   // online-softmax state of row 4*wave + lane/16 (owner lanes)
    const unsigned ks2_magic = (65536u + ks2 - 1) / ks2, ks_magic = (65536u + p.ks - 1) / p.ks;

    HICOM_TR();   // prologue: tables written (this wave); barrier [A] of tile 0 publishes them

    // stream slot -> token index (clamped to the last valid slot of this workgroup)
    // (s / WSZ by multiply-shift: exact for s < 2^16, i.e. <= kMaxWinPerWg * 64 tokens per workgroup)
    auto token_of = [&](int s) -> long {
        s = s < total ? s : total - 1;
        int wr = (int)(((unsigned)s * wsz_magic) >> 16);
        int i = s - wr * p.WSZ;
        if (i < 0) { i += p.WSZ; wr -= 1; }
        return (long)worg[wr] + win_off[i];
    };

    auto stage = [&](int tile, int buf) {
        const int row = 4 * wave + (lane >> 4), cpos = lane & 15;      // pieces of wave w cover rows 4w..4w+3
        const long tok = token_of(tile * 16 + row);
        const char* src = reinterpret_cast<const char*>(p.ff) + tok * (long)(E * 2) + 16 * (cpos ^ fswz(row));
#pragma unroll
        for (int i = 0; i < PIECES / 4; ++i) {
            const int pi = wave + 4 * i;                                // pi & 3 == wave, blk = i
            char* dst = tilebuf + buf * TILE_BYTES + pi * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                             (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
        }
    };

    const int orow = 4 * wave + (lane >> 4), tk = lane & 15;   // softmax ownership: (row, token slot) of this lane
    const int q4 = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 4 * fsig(kg) + q4;
    const int rd_row_off = r16 * 256, rd_swz = fswz(r16);
    const int tr_row_off = trow * 256 + 8 * (pp & 1), tr_swz = fswz(trow);

    // frames_embed B fragments (token slot r16, this wave's channel slice), fetched ONE TILE AHEAD
    // into the registers the previous tile's local-score MFMAs have just released
    auto load_fe = [&](int tile) {
        const uint16_t* src = p.fe + token_of(tile * 16 + r16) * E + SLICE * wave + 8 * kg;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) bfe[s] = *reinterpret_cast<const bf16x8*>(src + 32 * s);
    };

    for (int tile = 0; tile < ntile; ++tile) {
        const int cur = tile & 1;
        const int s0 = tile * 16;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // [A]
        HICOM_TR();   // tile: past [A] (data landed)
        const char* img = tilebuf + cur * TILE_BYTES;

        // window bookkeeping of this tile and the score-side pos-emb of this lane's (row, token slot): its
        // two dependent LDS lookups are issued here, under the score MFMAs, not in the serial softmax section
        int wr0 = (int)(((unsigned)s0 * wsz_magic) >> 16), i0 = s0 - wr0 * p.WSZ;
        if (i0 < 0) { i0 += p.WSZ; wr0 -= 1; }
        // local rows of the (at most two) windows this tile touches -- wave-uniform
        const int rowA = R + (wb + wr0) % NLOC;
        const int rowB = (rowA + 1 < 16) ? rowA + 1 : R;
        int oi = i0 + tk, owr = wr0;
        if (oi >= p.WSZ) { oi -= p.WSZ; owr += 1; }
        const bool in = s0 + tk < total;
        float posb = 0.f;
        if (p.pos_a && orow < R) {
            const int txy = win_txy[oi], base = wtxy[in ? owr : 0];
            const int f = (base >> 16) + (txy >> 16), y = ((base >> 8) & 255) + ((txy >> 8) & 255), x = (base & 255) + (txy & 255);
            int orow_c = orow;                           // opaque: recomputed per tile instead of held in a register
            asm volatile("" : "+v"(orow_c));
            const float* ap = a_pos + orow_c * (kMaxFramesPerWg + p.H + p.W);
            posb = ap[f] + ap[kMaxFramesPerWg + y] + ap[kMaxFramesPerWg + p.H + x];
        }

        // ---- local logits (rows >= R) from the frames_embed fragments (landed with [A]) -------------
        // These MFMAs come BEFORE the next tile's requests are issued: hipcc guards the use of `bfe` with
        // s_waitcnt vmcnt(0), and placed after the LDS-DMA issue that wait would drain the prefetch it
        // has just started (zero lookahead).  Here nothing is outstanding yet, so it is free.
        // Three accumulators: a single one would serialise the 9 MFMAs on their own latency.
        f32x4 e0 = f32x4{0.f, 0.f, 0.f, 0.f}, e1 = e0, e2 = e0;
#pragma unroll
        for (int s = 0; s < KSTEPS; s += 3) {
            e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[s], bfe[s], e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[s + 1], bfe[s + 1], e1, 0, 0, 0);
            e2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[s + 2], bfe[s + 2], e2, 0, 0, 0);
        }
        const f32x4 sfe = (e0 + e1) + e2;
        __builtin_amdgcn_sched_barrier(0);
        stage(tile + 1, cur ^ 1);                   // branch-free: past the end it re-requests the last tokens (clamped)

        // ---- global logits (rows < R) from the frames_feature tile in LDS -----------------------
        // All 9 fragment reads are in flight together (in the registers `bfe` has just released; its refill
        // is issued after these MFMAs), then 18 MFMAs on three accumulators: one LDS round trip per tile
        // instead of nine.
        bf16x8 bff[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int ch0 = SLICE * wave + 32 * s;
            const int blk = ch0 >> 7, cbase = (ch0 & 127) >> 3;
            bff[s] = *reinterpret_cast<const bf16x8*>(img + blk * 4096 + rd_row_off + 16 * ((cbase + kg) ^ rd_swz));
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 f0 = f32x4{0.f, 0.f, 0.f, 0.f}, f1 = f0;   // hi-plane and lo-plane chains, interleaved
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            f0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[s], bff[s], f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[s], bff[s], f1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the refill below the MFMAs that read the shared registers
        load_fe(tile + 1);                          // unconditional, so the registers are dead across the score phase
        const f32x4 sff = f0 + f1;

#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave * 256 + (4 * kg + j) * 16 + r16] = (4 * kg + j < R) ? sff[j] : sfe[j];
        lds_barrier();                                                 // [B]
        HICOM_TR();   // tile: past [B] (scores done)

        // ---- softmax: each wave owns 4 of the 16 rows; lane = (row 4*wave + lane/16, token slot lane%16).
        // (One logit per lane instead of four replicated in every wave; P, alpha and l are shared
        // through 1.2 KB of LDS.)
        {
            const int row = orow;
            float lgt = (red[row * 16 + tk] + red[256 + row * 16 + tk]) + (red[512 + row * 16 + tk] + red[768 + row * 16 + tk]);
            bool valid;
            if (row < R) {
                valid = in;
                lgt += posb;
            } else {
                valid = in && (row == (owr == wr0 ? rowA : rowB));
                lgt = lgt * p.l_scale + p.l_bias;
            }
            const float tmax = row16_max(valid ? lgt : -1.0e30f);
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = fast_exp(m_run - m_new);
            const float pr = valid ? fast_exp(lgt - m_new) : 0.f;
            l_run = l_run * alpha + row16_sum(pr);
            m_run = m_new;
            uint16_t h, l;
            split_bf16(pr, h, l);
            p_hi[row * kPStride + tk] = h;
            p_lo[row * kPStride + tk] = l;
            if (tk == 0) {
                alpha_s[row] = alpha;
                lrun_s[row] = l_run;
            }
        }
        lds_barrier();                                                 // [C] P / alpha / l visible to every wave
        HICOM_TR();   // tile: past [C] (softmax done)

        const bf16x4 phi = *reinterpret_cast<const bf16x4*>(p_hi + r16 * kPStride + 4 * fsig(kg));
        const bf16x4 plo = *reinterpret_cast<const bf16x4*>(p_lo + r16 * kPStride + 4 * fsig(kg));
        const f32x4 al = *reinterpret_cast<const f32x4*>(alpha_s + 4 * kg);
        if (__any(al[0] != 1.0f || al[1] != 1.0f || al[2] != 1.0f || al[3] != 1.0f)) {
#pragma unroll
            for (int cb = 0; cb < CBLK; ++cb) {
                acc[cb][0] *= al[0]; acc[cb][1] *= al[1]; acc[cb][2] *= al[2]; acc[cb][3] *= al[3];
            }
        }
        // ---- t / y / x marginals of the global weights, per window, as ONE more 16x16 MFMA tile:
        // MG += P . onehot(bin of each token).  A tile touches at most two consecutive windows: wave 0
        // accumulates the even-numbered one, wave 1 the odd one.  Rescaled by alpha like ACC, so a window's
        // bins end up relative to the running max at its last tile, which is stored next to them.  The
        // accumulators stay in LDS (the register file is full); each is touched by one wave only.
        // (LDS float atomics were measured at ~4 clk per lane: 3x slower than this.)
        if (p.part_marg && wave < 2) {
            bf16x4 bm;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int i = i0 + 4 * fsig(kg) + u;
                const int second = i >= p.WSZ ? 1 : 0;
                i -= second * p.WSZ;
                const int t2 = (int)(((unsigned)i * ks2_magic) >> 16), rem = i - t2 * ks2;
                const int h2 = (int)(((unsigned)rem * ks_magic) >> 16), w2 = rem - h2 * p.ks;
                const bool hit = r16 == t2 || r16 == p.kt + h2 || r16 == p.kt + p.ks + w2;
                bm[u] = (hit && ((wr0 + second) & 1) == wave) ? (short)0x3F80 : (short)0;
            }
            f32x4 mg = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(phi, bm, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            mg = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(plo, bm, mg, 0, 0, 0);
            int lane_m = lane;                           // opaque: this block's addresses are not loop invariants
            asm volatile("" : "+v"(lane_m));
            const int r16m = lane_m & 15, kgm = lane_m >> 4;
            if (r16m < kMargW) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * kgm + j < R) {
                        float* a = mbin + (wave * R + 4 * kgm + j) * kMargW + r16m;
                        *a = fmaf(*a, al[j], mg[j]);
                    }
            }
        }
        // ACC += P . x.  The transposed LDS reads are issued as inline asm with our own lgkmcnt wait:
        // through the builtin, hipcc orders them behind ALL outstanding vector-memory traffic
        // (s_waitcnt vmcnt(0)), which would drain the next tile's LDS-DMA and frames_embed prefetch.
        {
            const unsigned img_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(img);
            constexpr int G = 3;                     // reads in flight per group (register budget: 256 VGPRs)
            static_assert(CBLK % G == 0, "column blocks per group");
#pragma unroll
            for (int g0 = 0; g0 < CBLK; g0 += G) {
                bf16x4 bv[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int ch0 = SLICE * wave + 16 * (g0 + u);
                    const int blk = ch0 >> 7, c2 = (ch0 & 127) >> 3;
                    const unsigned addr = img_lds + blk * 4096 + tr_row_off + 16 * ((c2 + (pp >> 1)) ^ tr_swz);
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bv[u]) : "v"(addr));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    acc[g0 + u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(phi, bv[u], acc[g0 + u], 0, 0, 0);
                    acc[g0 + u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(plo, bv[u], acc[g0 + u], 0, 0, 0);
                }
            }
        }

        HICOM_TR();   // tile: P.x issued
        // ---- a window completed in this tile: emit its local context, recycle its row ------------
        if (i0 + 16 >= p.WSZ) {
            const int w = wb + wr0;
            const int row = rowA;                          // wave-uniform
            const float linv = 1.0f / lrun_s[row];
            const int rk = row >> 2, rj = row & 3;
            // The row sits in 16 lanes as 18 values 16 channels apart: stored from there it would be 36
            // 2-byte store instructions per wave (each a full pass of the address unit).  Two hops through a
            // wave-private scratch inside red[1..3] (idle between [C] and the next [A]) regroup it so that
            // 36 lanes hold 4 consecutive channels each: 2 wide stores per plane instead.
            int lane_c = lane;                           // opaque copy: keeps this block's address math out of the loop-invariant registers
            asm volatile("" : "+v"(lane_c));
            float* wsc = red + 256 + wave * (SLICE / 2);
            static_assert(CBLK % 2 == 0 && 4 * (SLICE / 2) <= 3 * 256 && SLICE / 8 <= 64, "scratch of the row regroup");
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j == rj) {
                        if (kg == rk) {
#pragma unroll
                            for (int c = 0; c < CBLK / 2; ++c) {
                                wsc[16 * c + r16] = acc[half * (CBLK / 2) + c][j] * linv;
                                acc[half * (CBLK / 2) + c][j] = 0.f;
                            }
                        }
                    }
                }
                if (lane_c < SLICE / 8) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wsc + 4 * lane_c);
                    const long o = (long)w * E + SLICE * wave + half * (SLICE / 2) + 4 * lane_c;
                    if (p.ctx_local) *reinterpret_cast<f32x4*>(p.ctx_local + o) = v;
                    if (p.ctx_hi) {
                        uint16_t h[4], l[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) split_bf16(v[u], h[u], l[u]);
                        *reinterpret_cast<uint2*>(p.ctx_hi + o) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                        *reinterpret_cast<uint2*>(p.ctx_lo + o) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
                    }
                }
            }
            if (p.part_marg) {
                // the completed window's marginals (wave 0) and its reference max (owner lanes of the rows)
                float* o = p.part_marg + ((long)part * R * p.wpw + wr0) * kMargW;
                const int r16c = lane_c & 15, kgc = lane_c >> 4;
                if (wave == (wr0 & 1) && r16c < kMargW - 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * kgc + j < R) {
                            float* a = mbin + (wave * R + 4 * kgc + j) * kMargW + r16c;
                            o[(4 * kgc + j) * p.wpw * kMargW + r16c] = *a;
                            *a = 0.f;
                        }
                }
                const int orow2 = 4 * wave + kgc;
                if (orow2 < R && r16c == kMargW - 1) o[orow2 * p.wpw * kMargW + kMargW - 1] = m_run;
            }
            if (4 * wave + (lane >> 4) == row) { m_run = -1.0e30f; l_run = 0.f; }   // owner lanes recycle the row
        }
        HICOM_TR();   // tile: window completion (if any) issued
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped requests of the last iteration
    HICOM_TR();   // loop done
    // ---- partial global state of this workgroup --------------------------------------------------
    const long prow = (long)part * 16;
    {
        const int row = 4 * wave + (lane >> 4);
        if ((lane & 15) == 0 && row < R) {
            p.part_m[prow + row] = m_run;
            p.part_l[prow + row] = l_run;
        }
    }
    // The accumulator rows go out through the (now idle) tile buffers, regrouped so that every lane stores
    // 16 contiguous bytes: 11 store instructions per wave instead of 72 four-byte ones.
    __syncthreads();                                   // every wave is done with the last tile image
    {
        float* est = reinterpret_cast<float*>(tilebuf) + wave * (R * SLICE);       // wave-private [R][SLICE]
#pragma unroll
        for (int cb = 0; cb < CBLK; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * kg + j < R) est[(4 * kg + j) * SLICE + 16 * cb + r16] = acc[cb][j];
        const int n4 = R * (SLICE / 4);
        for (int it = lane; it < n4; it += 64) {
            const int row = it / (SLICE / 4), c4 = it - row * (SLICE / 4);
            *reinterpret_cast<f32x4*>(p.part_acc + (prow + row) * E + SLICE * wave + 4 * c4) = *reinterpret_cast<const f32x4*>(est + 4 * it);
        }
    }
    HICOM_TR();   // epilogue issued
}

}  // namespace hicom

using namespace hicom;

static int fused_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

#ifdef HICOM_TRACE
extern "C" int hicom_debug_fused_trace(void* dst, int64_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(hicom::g_fused_trace), (size_t)bytes) == hipSuccess ? HICOM_OK : HICOM_ELAUNCH;
}
#endif

extern "C" int hicom_fused_stream_nparts(int32_t n_windows) {
    if (n_windows <= 0) return HICOM_EINVAL;
    const int slots = 2 * fused_num_cus();                       // 2 resident workgroups per CU
    int wpw = (n_windows + slots - 1) / slots;                   // equal windows per workgroup
    if (wpw > kMaxWinPerWg) wpw = kMaxWinPerWg;
    return (n_windows + wpw - 1) / wpw;
}

extern "C" int hicom_fused_stream_fwd(const void* ff, const void* fe, int32_t T, int32_t H, int32_t W, int32_t E,
                                      int32_t kt, int32_t ks, const void* q_hi, const void* q_lo, int32_t rows,
                                      float l_scale, float l_bias, const float* pos_a, int32_t pos_stride,
                                      int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                      float* part_m, float* part_l,
                                      float* part_acc, float* part_marg, int32_t nparts, float* ctx_local, void* ctx_hi,
                                      void* ctx_lo, void* stream) {
    HICOM_REQUIRE(ff && fe && q_hi && q_lo && part_m && part_l && part_acc && (part_marg || !pos_a), HICOM_EINVAL,
                  "fused_stream: NULL pointer");
    HICOM_REQUIRE(ctx_local || (ctx_hi && ctx_lo), HICOM_EINVAL, "fused_stream: no local output");
    HICOM_REQUIRE(E == 1152, HICOM_EUNSUP, "fused_stream: E=%d (only 1152)", E);
    HICOM_REQUIRE(T > 0 && H > 0 && W > 0 && kt > 0 && ks > 0 && T % kt == 0 && H % ks == 0 && W % ks == 0, HICOM_EUNSUP,
                  "fused_stream: windows must partition the [%d,%d,%d] grid exactly", T, H, W);
    const int wsz = kt * ks * ks;
    HICOM_REQUIRE(wsz >= 16 && wsz <= 64, HICOM_EUNSUP, "fused_stream: window of %d tokens (16..64 supported)", wsz);
    HICOM_REQUIRE(rows > 0 && rows <= 14, HICOM_EUNSUP, "fused_stream: %d global rows (<= 14: >= 2 local rows needed)", rows);
    HICOM_REQUIRE(H < 256 && W < 256 && (long)T * H * W < (1L << 31), HICOM_EUNSUP, "fused_stream: grid too large");
    const int NW = (T / kt) * (H / ks) * (W / ks);
    HICOM_REQUIRE(nparts > 0 && nparts <= NW, HICOM_EINVAL, "fused_stream: nparts");
    const int wpw = (NW + nparts - 1) / nparts;
    HICOM_REQUIRE(wpw <= kMaxWinPerWg && (long)(nparts - 1) * wpw < NW, HICOM_EINVAL,
                  "fused_stream: nparts=%d gives %d windows per workgroup (max %d, no empty workgroup)", nparts, wpw, kMaxWinPerWg);
    // frames a workgroup may touch: the t-groups its windows span
    const int per_t = (H / ks) * (W / ks);
    const int span = (wpw + per_t - 2) / per_t + 1;
    HICOM_REQUIRE(span * kt <= kMaxFramesPerWg, HICOM_EUNSUP, "fused_stream: a workgroup would span %d frames", span * kt);
    HICOM_REQUIRE(kt + 2 * ks <= kMargW - 1, HICOM_EUNSUP, "fused_stream: window %dx%dx%d exceeds the marginal bin width", kt, ks, ks);
    const size_t smem = 2 * 9 * 4096 + 4096 + (part_marg ? (size_t)2 * rows * kMargW * 4 : 0) + 128 +
                        (64 + 64 + 2 * kMaxWinPerWg) * 4 + (size_t)rows * (kMaxFramesPerWg + H + W) * 4;
    HICOM_REQUIRE(smem <= 81920, HICOM_EUNSUP, "fused_stream: H + W = %d does not fit the LDS budget", H + W);
    FusedParams p;
    p.ff = (const uint16_t*)ff; p.fe = (const uint16_t*)fe; p.T = T; p.H = H; p.W = W;
    p.kt = kt; p.ks = ks; p.nwy = H / ks; p.nwx = W / ks; p.NW = NW; p.WSZ = wsz;
    p.qhi = (const uint16_t*)q_hi; p.qlo = (const uint16_t*)q_lo; p.R = rows;
    p.l_scale = l_scale; p.l_bias = l_bias;
    p.pos_a = pos_a; p.pos_stride = pos_stride; p.t0i = t_index0; p.y0i = y_index0; p.x0i = x_index0;
    p.part_m = part_m; p.part_l = part_l; p.part_acc = part_acc; p.part_marg = part_marg; p.ctx_local = ctx_local; p.ctx_hi = (uint16_t*)ctx_hi; p.ctx_lo = (uint16_t*)ctx_lo; p.wpw = wpw;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(fused_stream_kernel<9>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        attr_set = true;
    }
    hipLaunchKernelGGL(fused_stream_kernel<9>, dim3((unsigned)nparts), dim3(256), smem, (hipStream_t)stream, p);
    return hicom_host::check_launch("fused_stream");
}
