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
// so the P.x product accumulates the global head contexts AND the open windows' local contexts in
// the same accumulator registers.
//
// Token order: a workgroup walks a contiguous range of windows in window-major order (the kt*ks*ks
// tokens of a window, then the next window), 16 tokens per tile.  At most 2 windows are open per
// tile, so NLOC = 16 - R >= 2 local rows suffice; when a window's last token has been accumulated its
// row is normalised, written out and recycled.
//
// Execution model: ONE 768-thread workgroup per CU with fixed roles.
//   4 LOADER waves  own every vector-memory request of the tile loop: both visual tensors travel
//                   HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a
//                   2 + 2 slot ring of swizzled tile images ([E/128][16 tokens][256 B]).  Issuing a DMA
//                   piece costs the issuing wave 60-185 cycles (MI355X_MICROARCH.md), which in the
//                   previous 4-wave design sat in front of the MFMAs of every tile; here it is paid by
//                   waves that have nothing else to do.  The loaders wait on their own counted vmcnt
//                   and publish a tile at barrier [A] (two barriers per tile in all); frames_embed of tile t+2 is requested as soon as
//                   the local scores of tile t have consumed its slot (barrier [B]), frames_feature of
//                   tile t+1 when the P.x of tile t-1 has released its slot (barrier [A]).
//   8 COMPUTE waves each own a 144-channel slice: score B fragments by ds_read_b64 (rows), P.x B
//                   fragments by ds_read_b64_tr_b16 (columns) from the same image, v_mfma_f32_16x16x16_bf16
//                   throughout, fp32 online softmax (2 rows per wave, DPP row reductions).
// The positional marginals of the global weights (value-side pos-emb) are one more 16x16 MFMA tile
// per tile (P x one-hot bins), kept in LDS and written out per window.
#include <stdlib.h>

#include "common.hpp"

namespace hicom {

// Dev-only phase timeline (tools/fused_trace.py builds a second library with -DHICOM_TRACE): lane 0 of
// compute waves 0 / 2 and of loader 0 stamp s_memtime at the phase boundaries.  Compiled out of the product.
#ifdef HICOM_TRACE
__device__ unsigned long long g_fused_trace[1024 * 3 * 256];
#define HICOM_TR(who) do { if (lane == 0 && wave == (who == 0 ? 0 : who == 1 ? 2 : kRingC) && tr_n < 256) g_fused_trace[(blockIdx.x * 3 + who) * 256 + tr_n] = __builtin_readcyclecounter(); ++tr_n; } while (0)
#else
#define HICOM_TR(who) do {} while (0)
#endif

// cache policy of the token stream's LDS-DMA loads: every token is read exactly once per call, so the loads are
// non-temporal (aux = 2: MI355X_MICROARCH.md "nt-weights": streamed once-read bytes) and leave L2 / Infinity Cache to the
// weights and partial states of the kernels that follow
#ifndef HICOM_RING_AUX
#define HICOM_RING_AUX 2
#endif
constexpr int kRingC = 8;                 // compute waves
constexpr int kRingL = 4;                 // loader waves
constexpr int kRingThreads = 64 * (kRingC + kRingL);
constexpr int kMaxWinPerWg = 32;
constexpr int kMaxFramesPerWg = 8;
constexpr int kPosPerThread = 2;          // score-side pos-emb entries fetched per compute thread: rows * (8 + H + W) <= 1024
constexpr int kPStride = 32;              // halfwords between rows of the softmax-weight planes (they live in red[0]'s rows)
constexpr int kRegroup = 80;               // channels per hop of the completed-row regroup (multiple of 16)

struct RingParams {
    const uint16_t* ff;
    const uint16_t* fe;    // frames_embed (fused_ring_kernel) ...
    const float* llog;     // ... or precomputed local logits fe_n . guide, one per token [T*H*W] (fused_ring_logits_kernel)
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
    float* part_acc;       // [nparts][16][E], rows < R: un-normalised fp32 accumulators ...
    _Float16* part_ctx16;  // ... or (not NULL) the NORMALISED partial contexts acc / l as one fp16 plane, same shape
    const uint16_t* pe_hi; // [P][E] bf16 hi / lo planes of the per-axis sinusoid tables (value-side pos-emb), or NULL
    const uint16_t* pe_lo;
    float* ctx_local;      // [NW][E] fp32 window contexts (may be NULL)
    uint16_t* ctx_hi;      // [NW][E] the same as bf16 hi / lo planes for hicom_planes_gemm_fwd (may be NULL)
    uint16_t* ctx_lo;
    _Float16* ctx_f16;     // [NW][E] the same as ONE fp16 plane (saturating) for hicom_readout16_gemm_fwd (may be NULL)
    int wpw;               // windows per workgroup
    unsigned long long* zero_ptr;   // scratch cleared for the launches behind this one (workgroup 0), or NULL
    int zero_n;            // ... 8-byte words
    // Round 6: value-side pos-emb OUTSIDE the ring.  With part_marg set (and pe_hi NULL) the kernel does not multiply its marginals by
    // the pe rows behind the token stream (2 slot tiles x 2 planes = ~184 KB of pe rows per CU through the ring's barrier cadence,
    // 5.5k clocks with HBM idle); it writes them out instead -- NORMALISED like part_ctx16 (marginal / l), fp16, rows < R, in ABSOLUTE
    // slot order [T frames | H grid rows | W grid columns | zero padding to marg_slots] -- and the merge multiplies the merged
    // marginals by v_proj . pe^T, a weight-only table (merge_item.hpp).
    _Float16* part_marg;   // [nparts][R][marg_slots], or NULL
    int marg_slots;
};

// Position of 16-byte chunk c of image row r inside the row: c ^ fswz(r).  The row -> XOR map is chosen so that
//  * the score fragment reads (ds_read_b128; the hardware services lanes in four 16-lane groups, each mixing the
//    rows {0-3, 12-15} at chunk c with the rows {4-11} at chunk c ^ 1, c even) hit 16 distinct slots: both row
//    sets map to whole {2m, 2m+1} pairs; and
//  * the transposed reads (ds_read_b64_tr_b16; 32-lane groups, rows {0-3, 8-11} or {4-7, 12-15}, chunks c and
//    c ^ 1) hit 32 distinct bank pairs: the 8 rows of a group fall into 8 different pairs.
// rows 0-3 -> 0,2,4,6   4-7 -> 8,10,12,14   8-11 -> 9,11,13,15   12-15 -> 1,3,5,7
__device__ __forceinline__ int fswz(int r) { return ((r & 3) << 1) | ((r >> 3) & 1) | ((((r >> 2) ^ (r >> 3)) & 1) << 3); }
__device__ __forceinline__ int fsig(int g) { return ((g & 1) << 1) | (g >> 1); }
// float4 group of (row, k-group) inside a row of the logit exchange: rows 8-15 store their groups in reverse
// pairing so that the four 16-lane service groups of a ds_read/write_b128 hit 16 distinct slots
__device__ __forceinline__ int fxg(int row, int kg) { return kg ^ (((row >> 3) & 1) * 3); }

template <int N>
__device__ __forceinline__ void ring_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LOGITS = false: both visual tensors are streamed (frames_embed for the local scores);  LOGITS = true: the local logits come
// precomputed (RingParams::llog: SURVEY.md §8 row f2 -- the producer of frames_embed, the SigLIP head projection of reference
// encoder.py:284-286, hands over fe_n . guide per token from its GEMM epilogue, so frames_embed is never written or read) and the
// ring's four slots all carry frames_feature: everything of tile t+3 is requested right behind barrier [B] of tile t, the compute
// waves fetch the 4 logits of their token slots one tile ahead with ordinary loads, no local score MFMAs.  ONE kernel body: the
// two forms differ in the loader schedule, the score operand and the cadence of the pos-emb images behind the stream.
template <int NB, bool LOGITS>
__global__ __launch_bounds__(kRingThreads, 1) void fused_ring_kernel(const uint16_t* ff0, const uint16_t* fe0, int H0, int W0, int kt0, int ks0, int nwy0, int nwx0, int NW0, int WSZ0, int wpw0,
                                                                     RingParams p_in) {
    // (the leading scalars repeat what the loaders' FIRST requests depend on -- stream bases and the window geometry: preloaded into SGPRs at
    // wave launch, build_native.py -amdgpu-kernarg-preload-count; the rest of the argument block is fetched under the prologue as before)
    RingParams p = p_in;
    p.ff = ff0; p.fe = fe0; p.H = H0; p.W = W0; p.kt = kt0; p.ks = ks0; p.nwy = nwy0; p.nwx = nwx0; p.NW = NW0; p.WSZ = WSZ0; p.wpw = wpw0;
#ifdef HICOM_TRACE
    const unsigned long long tr_entry = __builtin_readcyclecounter();
    const unsigned long long tr_entry_rt = __builtin_amdgcn_s_memrealtime();      // 100 MHz, chip-wide: workgroups are comparable
#endif
    constexpr int E = NB * 128;
    constexpr int SLICE = E / kRingC;              // channels per compute wave
    constexpr int KS = SLICE / 16;                 // 16-column blocks of the P.x product per wave
    constexpr int K32 = SLICE / 32;                // score K-steps of 32 channels
    constexpr bool KTAIL = (SLICE % 32) != 0;      // + one K-step of 16 channels
    constexpr int TILE_BYTES = NB * 4096;          // one 16-token image
    constexpr int PIECES = NB * 4;                 // 1-KiB DMA pieces per image: (128-channel block, 4-token row group)
    constexpr int PPL = PIECES / kRingL;           // pieces per loader wave and image
    constexpr int NSLOT = 4;                       // LOGITS: ring slots (one tensor); the two-tensor form splits them 2 + 2
    static_assert(3 * PPL <= 63, "vmcnt is a 6-bit counter");
    static_assert(E % (16 * kRingC) == 0 && PIECES % kRingL == 0 && kRingL == 4, "slice / piece split");
    static_assert(SLICE % 16 == 0 && kRegroup % 16 == 0 && kRegroup / 4 <= 64, "row regroup: hops of kRegroup channels, one float4 per lane");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ffbuf = smem;                                                // [2][TILE_BYTES] frames_feature ring (LOGITS: [4])
    char* febuf = smem + 2 * TILE_BYTES;                               // [2][TILE_BYTES] frames_embed ring (LOGITS: slots 2, 3 of ffbuf)
    float* red = reinterpret_cast<float*>(smem + 4 * TILE_BYTES);     // [kRingC + 1][16][16] logit partials (channel slices) + the score-side pos-emb
    float* cscr = red + (kRingC + 1) * 256;                                  // [kRingC][kRegroup] wave-private regroup scratch of a completed row
    float* alpha_s = cscr + kRingC * kRegroup;                            // [kRingC][16] wave-private rescale factors in accumulator-row order
    int* win_txy = reinterpret_cast<int*>(alpha_s + kRingC * 16);   // [64] packed in-window coords (t2 << 16 | h2 << 8 | w2)
    int* wtxy = win_txy + 64;                                          // [kMaxWinPerWg] packed window base coords (frame offset << 16 | y0 << 8 | x0)
    int* slot_row = wtxy + kMaxWinPerWg;                               // [64] pe row of a compact pos-emb slot; [64] = number of slots
    unsigned char* ymap = reinterpret_cast<unsigned char*>(slot_row + 65);   // [64] grid row -> compact slot (255: not touched by this workgroup)
    unsigned char* xmap = ymap + 64;                                   // [64] grid column -> compact slot
    int* tokslot = reinterpret_cast<int*>(xmap + 64);                  // [16] compact pos-emb slots (frame | row << 8 | column << 16) of this tile's tokens
    int* wbase = tokslot + 16;                                         // LOGITS: [kMaxWinPerWg] token index of each window's first token
    int* win_off = wbase + (LOGITS ? kMaxWinPerWg : 0);                // LOGITS: [64] token-index offset of in-window position i
    float* a_pos = reinterpret_cast<float*>(win_off + (LOGITS ? 64 : 0));   // [R][kMaxFramesPerWg | H | W] score-side pos-emb per row

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tr_n = 0; (void)tr_n;
    const int part = blockIdx.x;
    const int wb = part * p.wpw;
    const int we = min(p.NW, wb + p.wpw);
    const int nwin = we - wb;
    const int total = nwin * p.WSZ;                 // tokens of this workgroup's stream
    const int ntile = (total + 15) >> 4;
    const int R = p.R, NLOC = 16 - R;
    const int ks2 = p.ks * p.ks, per_t = p.nwy * p.nwx;
    const unsigned wsz_magic = (65536u + p.WSZ - 1) / p.WSZ;    // s / WSZ by multiply-shift: exact for s < 2^16

    // =========================================================================================
    // LOADER waves
    // =========================================================================================
    if constexpr (LOGITS) {
        if (wave >= kRingC) {
            const int l = wave - kRingC;
            const int row = 4 * l + (lane >> 4), cpos = lane & 15;          // token slot of this lane, 16-byte chunk in the row
            const int lane_off = 16 * (cpos ^ fswz(row));
            // stream slot -> token index (slots past the end of the last tile re-read its last token).  Prologue form: plain
            // index arithmetic (five runtime divisions, ~120 instructions); the tile loop uses the two tables the compute waves
            // build before barrier [P] (window base + in-window offset: two LDS reads).
            auto tok_slow = [&](int s) -> long {
                s = s < total ? s : total - 1;
                int wr = (int)(((unsigned)s * wsz_magic) >> 16);
                int i = s - wr * p.WSZ;
                if (i < 0) { i += p.WSZ; wr -= 1; }
                const int w = wb + wr;
                const int t1 = w / per_t, r = w - t1 * per_t, h1 = r / p.nwx, w1 = r - h1 * p.nwx;
                const int t2 = i / ks2, ri = i - t2 * ks2, h2 = ri / p.ks, w2 = ri - h2 * p.ks;
                return ((long)(t1 * p.kt + t2) * p.H + (h1 * p.ks + h2)) * p.W + (w1 * p.ks + w2);
            };
            auto tok_fast = [&](int s) -> long {
                s = s < total ? s : total - 1;
                int wr = (int)(((unsigned)s * wsz_magic) >> 16);
                int i = s - wr * p.WSZ;
                if (i < 0) { i += p.WSZ; wr -= 1; }
                return (long)(wbase[wr] + win_off[i]);
            };
            auto issue = [&](const uint16_t* base, long off, char* img) {
                const char* src = reinterpret_cast<const char*>(base) + off;
    #pragma unroll
                for (int i = 0; i < PPL; ++i)                               // piece (block i, row group l)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                                     (__attribute__((address_space(3))) void*)(img + (4 * i + l) * 1024), 16, 0, HICOM_RING_AUX);
            };
            // Past the token stream the ring carries the value-side pos-emb: images ntile + 2 b (hi plane) and ntile + 2 b + 1
            // (lo plane) hold the pe rows of compact slot tile b, so the request cadence, the slots and the counted waits stay
            // exactly those of the stream, and the first pe images are in flight while the last token tiles are consumed.
            int ntot = ntile, nsl = 0;
            // byte offset of this lane's source row for ring image `img`
            auto off_of = [&](int img, auto tok_of) -> long {
                if (img < ntile) return tok_of(img * 16 + row) * (long)(E * 2) + lane_off;
                const int s = 16 * ((img - ntile) >> 1) + row;              // compact pos-emb slot
                return (long)slot_row[s < nsl ? s : 0] * (long)(E * 2) + lane_off;
            };
            auto img_base = [&](int img) -> const uint16_t* { return img < ntile ? p.ff : (((img - ntile) & 1) ? p.pe_lo : p.pe_hi); };
            // Request order (vmcnt retires in issue order): ff(0) ff(1) ff(2) | [P] | iteration t: ff(t+3).  At the top of
            // iteration t image t has to be complete; images t+1 and t+2 may stay in flight.
            auto wait_allow = [&](int n) {                                  // at most n of the youngest operations still in flight
                if (n >= 2 * PPL) ring_wait_vm<2 * PPL>();
                else if (n >= PPL) ring_wait_vm<PPL>();
                else ring_wait_vm<0>();
            };
            issue(p.ff, off_of(0, tok_slow), ffbuf);
            if (ntile > 1) issue(p.ff, off_of(1, tok_slow), ffbuf + TILE_BYTES);
            if (ntile > 2) issue(p.ff, off_of(2, tok_slow), ffbuf + 2 * TILE_BYTES);
            HICOM_TR(2);   // prologue requests issued
            __builtin_amdgcn_s_barrier();                                  // [P] (the compute waves' tables)
            if (p.pe_hi) {
                nsl = slot_row[64];
                ntot = ntile + 2 * ((nsl + 15) >> 4);
            }
            // images 1 and 2 when they are pe images (a stream of one or two tiles): their rows are tabled only now
            if (ntile <= 1 && ntot > 1) issue(img_base(1), off_of(1, tok_fast), ffbuf + TILE_BYTES);
            if (ntile <= 2 && ntot > 2) issue(img_base(2), off_of(2, tok_fast), ffbuf + 2 * TILE_BYTES);
            for (int t = 0; t < ntot; ++t) {
                wait_allow((t + 1 < ntot ? PPL : 0) + (t + 2 < ntot ? PPL : 0));
                HICOM_TR(2);   // tile: data landed
                __builtin_amdgcn_s_barrier();                              // [A] image t published; the slot of image t-1 released
                if (t == ntile) __builtin_amdgcn_s_barrier();              // [A'] (the compute waves table their marginals)
                HICOM_TR(2);   // tile: past [A]
                if ((p.pe_hi || p.part_marg) && l == 0 && t < ntile && lane < 16) {
                    // compact pos-emb slots of this tile's 16 tokens, for the marginal MFMA of the compute waves (read
                    // after [B]): the table walk costs a loader lane nothing that matters
                    const int s = t * 16 + lane;
                    int wr = (int)(((unsigned)s * wsz_magic) >> 16), i = s - wr * p.WSZ;
                    if (i < 0) { i += p.WSZ; wr -= 1; }
                    int v = 0xFFFFFF;
                    if (s < total) {
                        const int txy = win_txy[i], base = wtxy[wr];
                        const int f = (base >> 16) + (txy >> 16), y = ((base >> 8) & 255) + ((txy >> 8) & 255), x = (base & 255) + (txy & 255);
                        v = f | ((int)ymap[y] << 8) | ((int)xmap[x] << 16);
                    }
                    tokslot[lane] = v;
                }
                if (p.pos_a && l >= 1 && t < ntile) {
                    // score-side pos-emb of the R x 16 (row, token) pairs of this tile: the ninth "partial" of the
                    // logit exchange, tabled here (three loader waves) so that no compute wave walks the tables
                    const int q = (l - 1) * 64 + lane;                      // 0 .. 16 R - 1
                    if (q < 16 * R) {
                        const int prow = q >> 4, pos = q & 15;
                        const int slot = 4 * fsig(fxg(prow, pos >> 2)) + (pos & 3);   // token slot held at exchange position `pos`
                        const int s = t * 16 + slot;
                        int wr = (int)(((unsigned)s * wsz_magic) >> 16), i = s - wr * p.WSZ;
                        if (i < 0) { i += p.WSZ; wr -= 1; }
                        const int txy = win_txy[i], base = wtxy[s < total ? wr : 0];
                        const int f = (base >> 16) + (txy >> 16), y = ((base >> 8) & 255) + ((txy >> 8) & 255), x = (base & 255) + (txy & 255);
                        const float* ap = a_pos + prow * (kMaxFramesPerWg + p.H + p.W);
                        red[kRingC * 256 + q] = ap[f] + ap[kMaxFramesPerWg + y] + ap[kMaxFramesPerWg + p.H + x];
                    }
                }
                const long o_nx = t + 3 < ntot ? off_of(t + 3, tok_fast) : 0;   // address math ahead of the barrier
                lds_barrier();                                             // [B] tables of tile t visible
                HICOM_TR(2);   // tile: past [B]
                if (t + 3 < ntot) issue(img_base(t + 3), o_nx, ffbuf + ((t + 3) & (NSLOT - 1)) * TILE_BYTES);
                HICOM_TR(2);   // tile: next image requested
            }
            if (p.part_marg) {                                            // (the compute waves table and emit their marginals)
                __builtin_amdgcn_s_barrier();                              // [M1]
                __builtin_amdgcn_s_barrier();                              // [M2]
            }
            __builtin_amdgcn_s_barrier();                                  // [E] ring idle (nothing is in flight any more)
            return;
        }
    } else {
        if (wave >= kRingC) {
    #ifdef HICOM_TRACE
            if (lane == 0) g_fused_trace[(blockIdx.x * 3 + 2) * 256 + 240 + (wave - kRingC)] = __builtin_readcyclecounter();     // first instruction of loader l
    #endif
            const int l = wave - kRingC;
            const int row = 4 * l + (lane >> 4), cpos = lane & 15;          // token slot of this lane, 16-byte chunk in the row
            const int lane_off = 16 * (cpos ^ fswz(row));
            // stream slot -> byte offset of the token row (slots past the end of the last tile re-read its last token)
            auto src_off = [&](int tile) -> long {
                int s = tile * 16 + row;
                s = s < total ? s : total - 1;
                int wr = (int)(((unsigned)s * wsz_magic) >> 16);
                int i = s - wr * p.WSZ;
                if (i < 0) { i += p.WSZ; wr -= 1; }
                const int w = wb + wr;
                const int t1 = w / per_t, r = w - t1 * per_t, h1 = r / p.nwx, w1 = r - h1 * p.nwx;
                const int t2 = i / ks2, ri = i - t2 * ks2, h2 = ri / p.ks, w2 = ri - h2 * p.ks;
                const long tok = ((long)(t1 * p.kt + t2) * p.H + (h1 * p.ks + h2)) * p.W + (w1 * p.ks + w2);
                return tok * (long)(E * 2) + lane_off;
            };
            auto issue = [&](const uint16_t* base, long off, char* img) {
                const char* src = reinterpret_cast<const char*>(base) + off;
    #pragma unroll
                for (int i = 0; i < PPL; ++i)                               // piece (block i, row group l)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                                     (__attribute__((address_space(3))) void*)(img + (4 * i + l) * 1024), 16, 0, HICOM_RING_AUX);
            };
            // Past the token stream the ring carries the value-side pos-emb: "tile" ntile + b holds the pe rows of
            // compact slot tile b -- hi plane where frames_embed goes, lo plane where frames_feature goes -- so the
            // request cadence, the slots and the counted waits stay exactly those of the stream, and the first pe
            // images are already in flight while the last token tiles are being consumed.
            int ntot = ntile, nsl = 0;
            // byte offset of this lane's source row for image `tile`, computed ahead of the barrier it is issued after
            auto off_of = [&](int tile) -> long {
                if (tile < ntile) return src_off(tile);
                const int s = 16 * (tile - ntile) + row;                    // compact pos-emb slot
                return (long)slot_row[s < nsl ? s : 0] * (long)(E * 2) + lane_off;
            };
            const long o0 = src_off(0);
            issue(p.fe, o0, febuf);
            issue(p.ff, o0, ffbuf);
            // Both images of tile 1 go out up front as well (token tiles need no table and all four ring slots are free at the start):
            // 144 KB in flight per CU during the launch ramp instead of 72 (round 4: -0.9 us per launch, tools/gpu_ab.sh) -- but BEHIND
            // [P]: issuing an image costs a loader ~2.5k clocks (18 pieces at 100-185 clocks), and the compute waves, ready with their
            // tables at ~5k clocks, would wait at [P] for it (tools/fused_trace.py: loaders at [P] after 9.9k clocks with tile 1 in front)
            const bool pre1 = ntile > 1;
            const long o1 = pre1 ? src_off(1) : 0;
            HICOM_TR(2);   // prologue requests issued
            __builtin_amdgcn_s_barrier();                                  // [P] (the compute waves' tables)
            if (pre1) {
                issue(p.fe, o1, febuf + TILE_BYTES);
                issue(p.ff, o1, ffbuf + TILE_BYTES);
            }
            if (p.pe_hi) {
                nsl = slot_row[64];
                ntot = ntile + ((nsl + 15) >> 4);
            }
            long o_ff = ntot > 1 ? off_of(1) : 0;                          // offset of tile t+1 (fe, then ff)
            if (ntot > 1 && !pre1) issue(1 < ntile ? p.fe : p.pe_hi, o_ff, febuf + TILE_BYTES);
            for (int t = 0; t < ntot; ++t) {
                const long o_fe = t + 2 < ntot ? off_of(t + 2) : 0;         // address math ahead of the wait
                HICOM_TR(2);   // tile: addresses ready
                // in flight, oldest first: fe(t) | ff(t) | fe(t+1) (| ff(1) at t = 0 when tile 1 was requested up front): everything
                // but the youngest image (the two youngest) has to land
                if (t == 0 && pre1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * PPL) : "memory");
                else if (t + 1 < ntot) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PPL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                HICOM_TR(2);   // tile: data landed
                __builtin_amdgcn_s_barrier();                              // [A] tile t published; ff slot of tile t-1 released
                if (t == ntile) __builtin_amdgcn_s_barrier();              // [A'] (the compute waves table their marginals)
                HICOM_TR(2);   // tile: past [A]
                if (t + 1 < ntot && !(t == 0 && pre1)) issue(t + 1 < ntile ? p.ff : p.pe_lo, o_ff, ffbuf + ((t + 1) & 1) * TILE_BYTES);
                HICOM_TR(2);   // tile: ff issued
                if ((p.pe_hi || p.part_marg) && l == 0 && t < ntile && lane < 16) {
                    // compact pos-emb slots of this tile's 16 tokens, for the marginal MFMA of the compute waves (read
                    // after [B]): the table walk costs a loader lane nothing that matters
                    const int s = t * 16 + lane;
                    int wr = (int)(((unsigned)s * wsz_magic) >> 16), i = s - wr * p.WSZ;
                    if (i < 0) { i += p.WSZ; wr -= 1; }
                    int v = 0xFFFFFF;
                    if (s < total) {
                        const int txy = win_txy[i], base = wtxy[wr];
                        const int f = (base >> 16) + (txy >> 16), y = ((base >> 8) & 255) + ((txy >> 8) & 255), x = (base & 255) + (txy & 255);
                        v = f | ((int)ymap[y] << 8) | ((int)xmap[x] << 16);
                    }
                    tokslot[lane] = v;
                }
                if (p.pos_a && l >= 1 && t < ntile) {
                    // score-side pos-emb of the R x 16 (row, token) pairs of this tile: the ninth "partial" of the
                    // logit exchange, tabled here (three loader waves) so that no compute wave walks the tables
                    const int q = (l - 1) * 64 + lane;                      // 0 .. 16 R - 1
                    if (q < 16 * R) {
                        const int prow = q >> 4, pos = q & 15;
                        const int slot = 4 * fsig(fxg(prow, pos >> 2)) + (pos & 3);   // token slot held at exchange position `pos`
                        const int s = t * 16 + slot;
                        int wr = (int)(((unsigned)s * wsz_magic) >> 16), i = s - wr * p.WSZ;
                        if (i < 0) { i += p.WSZ; wr -= 1; }
                        const int txy = win_txy[i], base = wtxy[s < total ? wr : 0];
                        const int f = (base >> 16) + (txy >> 16), y = ((base >> 8) & 255) + ((txy >> 8) & 255), x = (base & 255) + (txy & 255);
                        const float* ap = a_pos + prow * (kMaxFramesPerWg + p.H + p.W);
                        red[kRingC * 256 + q] = ap[f] + ap[kMaxFramesPerWg + y] + ap[kMaxFramesPerWg + p.H + x];
                    }
                }
                lds_barrier();                                             // [B] fe slot of tile t released; the tables written above are visible
                HICOM_TR(2);   // tile: past [B]
                if (t + 2 < ntot) issue(t + 2 < ntile ? p.fe : p.pe_hi, o_fe, febuf + (t & 1) * TILE_BYTES);
                HICOM_TR(2);   // tile: fe issued
                o_ff = o_fe;
            }
            if (p.part_marg) {                                            // (the compute waves table and emit their marginals)
                __builtin_amdgcn_s_barrier();                              // [M1]
                __builtin_amdgcn_s_barrier();                              // [M2]
            }
            __builtin_amdgcn_s_barrier();                                  // [E] ring idle (nothing is in flight any more)
            return;
        }
    }

    // =========================================================================================
    // COMPUTE waves
    // =========================================================================================
    const int r16 = lane & 15, kg = lane >> 4;
    const int ctid = tid;                                              // compute threads are 0 .. 511

    // ---- score-side pos table of this workgroup: its (cold) gathers go out FIRST, their latency runs under the operand loads and the
    // table building below, and the values are parked in LDS at the end of the prologue (they were requested last and waited for at
    // once: 4.5k of the prologue's 8.4k clocks, tools/fused_trace.py) ------------------------------------------------------------
    const int t1_first = wb / per_t;
    float apv[kPosPerThread];
    if (p.pos_a) {
        const int S = kMaxFramesPerWg + p.H + p.W, n_all = R * S;
#pragma unroll
        for (int u = 0; u < kPosPerThread; ++u) {
            const int e = ctid + 64 * kRingC * u;
            apv[u] = 0.f;
            if (e < n_all) {
                const int r = e / S, c = e - r * S;
                const int t = t1_first * p.kt + c;
                const int col = c < kMaxFramesPerWg ? p.t0i + t
                                                    : (c < kMaxFramesPerWg + p.H ? p.y0i + (c - kMaxFramesPerWg) : p.x0i + (c - kMaxFramesPerWg - p.H));
                if (c >= kMaxFramesPerWg || t < p.T) apv[u] = p.pos_a[(long)r * p.pos_stride + col];
            }
        }
    }
    // ---- A operand (hi / lo) of this wave's channel slice: lane (query row r16, k group kg) -------
    // SLICE = K32 steps of 32 channels (v_mfma_f32_16x16x32_bf16, the full-rate instruction) + an optional
    // 16-channel tail (v_mfma_f32_16x16x16_bf16 runs at half the rate per flop).
    bf16x8 ahi[K32], alo[K32];
    bf16x4 ahi_t = bf16x4{0, 0, 0, 0}, alo_t = bf16x4{0, 0, 0, 0};
    {
        const long off = (long)r16 * E + SLICE * wave;
#pragma unroll
        for (int s = 0; s < K32; ++s) {
            ahi[s] = *reinterpret_cast<const bf16x8*>(p.qhi + off + 32 * s + 8 * kg);
            alo[s] = *reinterpret_cast<const bf16x8*>(p.qlo + off + 32 * s + 8 * kg);
        }
        if (KTAIL) {
            ahi_t = *reinterpret_cast<const bf16x4*>(p.qhi + off + 32 * K32 + 4 * kg);
            alo_t = *reinterpret_cast<const bf16x4*>(p.qlo + off + 32 * K32 + 4 * kg);
        }
    }
    if (part == 0 && p.zero_ptr)                                     // (fixed-point accumulators of the merge + v_proj launch behind us)
        for (int i = ctid; i < p.zero_n; i += 64 * kRingC) p.zero_ptr[i] = 0ull;

#ifdef HICOM_TRACE
    if (lane == 0 && wave == 0) g_fused_trace[(blockIdx.x * 3 + 0) * 256 + 241] = __builtin_readcyclecounter();
#endif
    // ---- per-workgroup tables ---------------------------------------------------------------------
    if (ctid < p.WSZ) {
        const int t2 = ctid / ks2, r = ctid - t2 * ks2, h2 = r / p.ks, w2 = r - h2 * p.ks;
        win_txy[ctid] = (t2 << 16) | (h2 << 8) | w2;
        if constexpr (LOGITS) win_off[ctid] = (t2 * p.H + h2) * p.W + w2;
    }
    if (ctid < nwin) {
        const int w = wb + ctid;
        const int t1 = w / per_t, r = w - t1 * per_t, h1 = r / p.nwx, w1 = r - h1 * p.nwx;
        wtxy[ctid] = (((t1 - t1_first) * p.kt) << 16) | ((h1 * p.ks) << 8) | (w1 * p.ks);
        if constexpr (LOGITS) wbase[ctid] = (t1 * p.kt * p.H + h1 * p.ks) * p.W + w1 * p.ks;
    }

#ifdef HICOM_TRACE
    if (lane == 0 && wave == 0) g_fused_trace[(blockIdx.x * 3 + 0) * 256 + 242] = __builtin_readcyclecounter();
#endif
    if ((p.pe_hi || p.part_marg) && wave == 0) {
        // Compact pos-emb slots of this workgroup: the 8 frames from its first frame group, then only the grid
        // rows and columns its windows touch (a few of the H + W): fewer pe rows to multiply after the stream.
        ymap[lane] = 0;
        xmap[lane] = 0;
        if (lane < nwin) {
            const int w = wb + lane;
            const int t1 = w / per_t, r = w - t1 * per_t, h1 = r / p.nwx, w1 = r - h1 * p.nwx;
            for (int j = 0; j < p.ks; ++j) {
                ymap[h1 * p.ks + j] = 1;
                xmap[w1 * p.ks + j] = 1;
            }
        }
        const bool yu = lane < p.H && ymap[lane] != 0, xu = lane < p.W && xmap[lane] != 0;
        const unsigned long long ym = __ballot(yu), xm = __ballot(xu), below = (1ull << lane) - 1ull;
        const int ny = __popcll(ym);
        const int cy = kMaxFramesPerWg + __popcll(ym & below), cx = kMaxFramesPerWg + ny + __popcll(xm & below);
        ymap[lane] = yu ? (unsigned char)cy : (unsigned char)255;
        xmap[lane] = xu ? (unsigned char)cx : (unsigned char)255;
        if (lane < kMaxFramesPerWg) slot_row[lane] = p.t0i + min(t1_first * p.kt + lane, p.T - 1);
        if (yu) slot_row[cy] = p.y0i + lane;
        if (xu) slot_row[cx] = p.x0i + lane;
        if (lane == 0) slot_row[64] = kMaxFramesPerWg + ny + __popcll(xm);
    }

#ifdef HICOM_TRACE
    if (lane == 0 && wave == 0) g_fused_trace[(blockIdx.x * 3 + 0) * 256 + 243] = __builtin_readcyclecounter();
#endif
    if (p.pos_a) {
        const int n_all = R * (kMaxFramesPerWg + p.H + p.W);
#pragma unroll
        for (int u = 0; u < kPosPerThread; ++u) {
            const int e = ctid + 64 * kRingC * u;
            if (e < n_all) a_pos[e] = apv[u];
        }
    }
    f32x4 acc[KS];
#pragma unroll
    for (int cb = 0; cb < KS; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Online-softmax state of query row r16, replicated in the four lanes (kg = 0..3) that hold its tokens
    // and in every compute wave: the scores are computed TRANSPOSED (S^T = x . q^T), which leaves each lane
    // with (row r16, token slots 4*sig(kg) .. +3) -- exactly the A-operand layout of P for the P.x MFMAs.
    // The softmax therefore runs in registers in all waves at once: no shared P, no third barrier.
    float m_run = -1.0e30f, l_run = 0.f;
    f32x4 mgacc = f32x4{0.f, 0.f, 0.f, 0.f};   // waves < nslot_tiles: marginals (rows 4 kg + j, slot 16 wave + r16) of the global weights

    const int q4 = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 4 * fsig(kg) + q4;
    const int prow16 = 4 * fsig(r16 >> 2) + (r16 & 3);                      // token slot read as A-row r16 of the score MFMA
    const int rd_row = prow16 * 256, rd_swz = fswz(prow16);                 // row read: chunk c of the row sits at position c ^ swz
    const int tr_row_off = trow * 256 + 8 * (pp & 1), tr_swz = fswz(trow);
    const int ch_base = SLICE * wave;
    const int ts0 = 4 * fsig(kg);                                            // first token slot of this lane
    float* ascr = alpha_s + 16 * wave;                                       // wave-private: alpha in accumulator-row order

    // one P.x step over a 16-row image: ACC[u] += A(16 rows x [hi | lo of 16 rows' weights]) . [x ; x] per 16-channel block
    constexpr int PG = (KS % 6 == 0 && KS > 9) ? 6 : KS;               // blocks whose fragments are in flight together
    auto px_step = [&](unsigned img_lds, const bf16x8& a_op, f32x4 (&accr)[KS]) {
#pragma unroll
        for (int g0 = 0; g0 < KS; g0 += PG) {
            bf16x4 bv[PG];
#pragma unroll
            for (int u = 0; u < PG; ++u) {
                const int ch0 = ch_base + 16 * (g0 + u);
                const unsigned addr = img_lds + (ch0 >> 7) * 4096 + tr_row_off + 16 * ((((ch0 & 127) >> 3) + (pp >> 1)) ^ tr_swz);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bv[u]) : "v"(addr));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PG; ++u) {
                const bf16x8 b2 = bf16x8{bv[u][0], bv[u][1], bv[u][2], bv[u][3], bv[u][0], bv[u][1], bv[u][2], bv[u][3]};
                accr[g0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_op, b2, accr[g0 + u], 0, 0, 0);
            }
        }
    };
    HICOM_TR(0); HICOM_TR(1);   // prologue done (this wave)
    lds_barrier();                                                     // [P] tables ready
    const int nslot_tiles = (p.pe_hi || p.part_marg) ? (__builtin_amdgcn_readfirstlane(slot_row[64]) + 15) >> 4 : 0;   // 16-slot tiles of the compact pos-emb slots
    // raw local logits of this lane's 4 token slots, fetched one tile ahead (slots past the stream re-read its last token)
    auto fetch_logits = [&](int tile) -> f32x4 {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int s = tile * 16 + ts0 + j;
            s = s < total ? s : total - 1;
            int wr = (int)(((unsigned)s * wsz_magic) >> 16), i = s - wr * p.WSZ;
            if (i < 0) { i += p.WSZ; wr -= 1; }
            v[j] = p.llog[wbase[wr] + win_off[i]];
        }
        return v;
    };
    f32x4 el_next = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LOGITS) el_next = fetch_logits(0);

    for (int tile = 0; tile < ntile; ++tile) {
        const int cur = tile & 1;
        const int s0 = tile * 16;
        HICOM_TR(0); HICOM_TR(1);   // tile: arrive [A]
        lds_barrier();                                                 // [A] images of this tile landed; red free
        HICOM_TR(0); HICOM_TR(1);   // tile: past [A]
        const char* ffimg = ffbuf + (LOGITS ? (tile & (NSLOT - 1)) : cur) * TILE_BYTES;
        const char* feimg = febuf + cur * TILE_BYTES;                   // (two-tensor form only)
        const f32x4 el = el_next;                                       // LOGITS: raw local logits of this lane's 4 tokens
        if constexpr (LOGITS) {
            if (tile + 1 < ntile) el_next = fetch_logits(tile + 1);   // lands under this tile's MFMAs
        }

        // window bookkeeping of this tile (wave-uniform)
        int wr0 = (int)(((unsigned)s0 * wsz_magic) >> 16), i0 = s0 - wr0 * p.WSZ;
        if (i0 < 0) { i0 += p.WSZ; wr0 -= 1; }
        const int rowA = R + (wb + wr0) % NLOC;                        // local rows of the (at most two) windows in this tile
        const int rowB = (rowA + 1 < 16) ? rowA + 1 : R;

        // ---- logits, transposed: A = token rows of both images, B = the query fragments ---------------
        // fragment reads in groups of KG K-steps (both images), each group's reads in flight together: bounds the
        // registers the fragments hold (fat slices) while keeping one LDS round trip per group
        constexpr int KG = (K32 % 3 == 0 && K32 > 4) ? 3 : K32;
        bf16x4 bfe_t = bf16x4{0, 0, 0, 0}, bff_t;
        if (KTAIL) {
            const int cc = ((ch_base + 32 * K32) >> 3) + (kg >> 1);
            const int off = (cc >> 4) * 4096 + rd_row + 16 * ((cc & 15) ^ rd_swz) + 8 * (kg & 1);
            if constexpr (!LOGITS) bfe_t = *reinterpret_cast<const bf16x4*>(feimg + off);
            bff_t = *reinterpret_cast<const bf16x4*>(ffimg + off);
        }
        f32x4 e0 = f32x4{0.f, 0.f, 0.f, 0.f}, f0 = e0, f1 = e0;
#pragma unroll
        for (int g0 = 0; g0 < K32; g0 += KG) {
            bf16x8 bfe[KG], bff[KG];
#pragma unroll
            for (int u = 0; u < KG; ++u) {
                const int cc = ((ch_base + 32 * (g0 + u)) >> 3) + kg;   // 16-byte chunk of this lane over the whole row
                const int off = (cc >> 4) * 4096 + rd_row + 16 * ((cc & 15) ^ rd_swz);
                if constexpr (!LOGITS) bfe[u] = *reinterpret_cast<const bf16x8*>(feimg + off);
                bff[u] = *reinterpret_cast<const bf16x8*>(ffimg + off);
            }
            if (KG < K32) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < KG; ++u) {
                if constexpr (!LOGITS) e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfe[u], ahi[g0 + u], e0, 0, 0, 0);
                f0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bff[u], ahi[g0 + u], f0, 0, 0, 0);
                f1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bff[u], alo[g0 + u], f1, 0, 0, 0);
            }
            if (KG < K32) __builtin_amdgcn_sched_barrier(0);
        }
        if (KTAIL) {
            if constexpr (!LOGITS) e0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bfe_t, ahi_t, e0, 0, 0, 0);
            f0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bff_t, ahi_t, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bff_t, alo_t, f1, 0, 0, 0);
        }
        // The accumulators are read by VALU right away.  Required software wait states, CDNA4 ISA §4.1 (data-hazard table,
        // "XDL write VGPR -> VALU read / write of that VGPR"; summarised in cdna_hip_programming.md §5.7 item 2): an 8-pass
        // XDL op (32x32x16) needs 12, the 4-pass 16x16x32 / 16x16x16 forms used here fewer.  hipcc inserts them from its
        // own table, but for the K = 16 tail (v_mfma_f32_16x16x16_bf16) its padding proved too short under load on
        // gfx950 / ROCm 7.2 (logits of single tiles read stale whenever a co-resident wave delayed the matrix pipe;
        // tools/dbg_async.py reproduces it on the unpadded build).  16 explicit states = the table's largest entry
        // for any shape in this file, independent of the compiler's model.
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        {
            const f32x4 part_logit = (LOGITS || r16 < R) ? (f0 + f1) : e0;
            *reinterpret_cast<f32x4*>(red + wave * 256 + r16 * 16 + 4 * fxg(r16, kg)) = part_logit;
        }
        HICOM_TR(0); HICOM_TR(1);   // tile: arrive [B]
        lds_barrier();                                                 // [B] channel-slice partials exchanged
        HICOM_TR(0); HICOM_TR(1);   // tile: past [B]

        // ---- softmax of (row r16, 4 token slots) in registers, identically in every wave ---------------
        f32x4 lg = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kRingC; ++k) lg += *reinterpret_cast<const f32x4*>(red + k * 256 + r16 * 16 + 4 * fxg(r16, kg));
        if (p.pos_a && r16 < R) lg += *reinterpret_cast<const f32x4*>(red + kRingC * 256 + r16 * 16 + 4 * fxg(r16, kg));
        float pr[4];
        float tmax = -1.0e30f;
        bool valid[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = s0 + ts0 + j < total;
            if (r16 < R) {
                valid[j] = in;
            } else {
                const int row_of = (i0 + ts0 + j >= p.WSZ) ? rowB : rowA;
                valid[j] = in && r16 == row_of;
                lg[j] = (LOGITS ? el[j] : lg[j]) * p.l_scale + p.l_bias;
            }
            tmax = fmaxf(tmax, valid[j] ? lg[j] : -1.0e30f);
        }
        tmax = xrow4_max(tmax);
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = fast_exp(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pr[j] = valid[j] ? fast_exp(lg[j] - m_new) : 0.f;
            psum += pr[j];
        }
        l_run = l_run * alpha + xrow4_sum(psum);
        m_run = m_new;
        // P as ONE K=32 A operand: k = 8 kg + u carries the hi plane of this lane's token u, k = 8 kg + 4 + u its
        // lo plane; the B operand then holds the lane's 4 transposed x values twice.  One full-rate
        // v_mfma_f32_16x16x32_bf16 per 16-channel block instead of two half-rate 16x16x16.
        bf16x8 pw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint16_t h, l;
            split_bf16(pr[j], h, l);
            pw[j] = (short)h;
            pw[4 + j] = (short)l;
        }
        // accumulator rows are indexed 4 kg + j: fetch their rescale factors through a wave-private LDS hop,
        // only in the (rare, after the first tiles) case that some running max moved
        f32x4 al = f32x4{1.f, 1.f, 1.f, 1.f};
        if (__any(alpha != 1.0f)) {
            if (kg == 0) ascr[r16] = alpha;
            al = *reinterpret_cast<const f32x4*>(ascr + 4 * kg);
#pragma unroll
            for (int cb = 0; cb < KS; ++cb) {
                acc[cb][0] *= al[0]; acc[cb][1] *= al[1]; acc[cb][2] *= al[2]; acc[cb][3] *= al[3];
            }
        }
        // ---- value-side pos-emb, part 1 (reference projector.py:57-101 with :176-179): the context of a row is
        // sum_n p_n (x_n + pe_t[t_n] + pe_y[y_n] + pe_x[x_n]).  The pe part only needs the t / y / x MARGINALS of
        // the weights: MG += P . onehot(slot of each token) over the slots  frame (relative to this workgroup's
        // first frame) | grid row | grid column -- one more MFMA per tile on the wave that owns the 16-slot
        // block.  Rescaled by alpha like ACC.  Part 2 (after the stream) multiplies MG by the pe rows.
        if (nslot_tiles && wave < nslot_tiles) {
            const int col = 16 * wave + r16;
            bf16x4 bm;
            const int4 tsl = *reinterpret_cast<const int4*>(tokslot + ts0);   // slots of this lane's 4 tokens (tabled by a loader wave)
            const int tsv[4] = {tsl.x, tsl.y, tsl.z, tsl.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool hit = col == (tsv[u] & 255) || col == ((tsv[u] >> 8) & 255) || col == (tsv[u] >> 16);
                bm[u] = hit ? (short)0x3F80 : (short)0;
            }
            const bf16x8 bm2 = bf16x8{bm[0], bm[1], bm[2], bm[3], bm[0], bm[1], bm[2], bm[3]};
            const f32x4 mg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw, bm2, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");             // MFMA -> VALU read, as above
#pragma unroll
            for (int j = 0; j < 4; ++j) mgacc[j] = fmaf(mgacc[j], al[j], mg[j]);
        }
        // ---- ACC += P . x: the transposed fragment reads in groups of PG blocks in flight, then their MFMAs --------
        {
            const unsigned img_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(ffimg);
            px_step(img_lds, pw, acc);
        }

        HICOM_TR(0); HICOM_TR(1);   // tile: P.x issued
        // ---- a window completed in this tile: emit its local context, recycle its row ------------
        if (i0 + 16 >= p.WSZ) {
            const int w = wb + wr0;
            const int row = __builtin_amdgcn_readfirstlane(rowA);
            const float linv = 1.0f / __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l_run), row));
            const int rk = row >> 2, rj = row & 3;
            // The row sits in 16 lanes as KS values 16 channels apart.  A hop through a wave-private scratch
            // inside red (idle between the softmax reads and the next [A]... see the barrier below) regroups it
            // so that SLICE/4 lanes hold 4 consecutive channels each: one 16-byte (fp32) or 8-byte (bf16 plane)
            // store per lane.
            int lane_c = lane;                           // opaque copy: keeps this block's address math out of the loop-invariant registers
            asm volatile("" : "+v"(lane_c));
            // (two hops of <= kRegroup channels: the scratch has to fit beside the ring)
            float* wsc = cscr + wave * kRegroup;
#pragma unroll
            for (int c0 = 0; c0 < KS; c0 += kRegroup / 16) {
                constexpr int NBLK = kRegroup / 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j == rj) {
                        if (kg == rk) {
#pragma unroll
                            for (int c = 0; c < NBLK; ++c)
                                if (c0 + c < KS) {
                                    wsc[16 * c + r16] = acc[c0 + c][j] * linv;
                                    acc[c0 + c][j] = 0.f;
                                }
                        }
                    }
                }
                const int nch = (KS - c0 < NBLK ? KS - c0 : NBLK) * 16;        // channels of this hop
                if (4 * lane_c < nch) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wsc + 4 * lane_c);
                    const long o = (long)w * E + ch_base + 16 * c0 + 4 * lane_c;
                    if (p.ctx_local) *reinterpret_cast<f32x4*>(p.ctx_local + o) = v;
                    if (p.ctx_hi) {
                        uint16_t h[4], l[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) split_bf16(v[u], h[u], l[u]);
                        *reinterpret_cast<uint2*>(p.ctx_hi + o) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                        *reinterpret_cast<uint2*>(p.ctx_lo + o) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
                    }
                    if (p.ctx_f16) {
                        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                        half4_t hv;
#pragma unroll
                        for (int u = 0; u < 4; ++u) hv[u] = (_Float16)fminf(fmaxf(v[u], -65504.f), 65504.f);
                        *reinterpret_cast<half4_t*>(p.ctx_f16 + o) = hv;
                    }
                }
            }
            if (r16 == row) { m_run = -1.0e30f; l_run = 0.f; }   // every copy of the row's state is recycled
        }
    }

    if (p.pe_hi) {
        // ---- value-side pos-emb, part 2: ACC += MG . pe, as P.x steps over the pe tiles that follow the token
        // stream in the ring.  MG (fp32, in the accumulator layout of the waves that own the slot blocks) is
        // tabled in LDS ([slot][row], in `red`) and re-read in the A-operand layout; hi + lo planes of MG in one
        // K = 32 operand against the hi plane (frames_embed slot), then the lo plane (frames_feature slot).
        if constexpr (LOGITS) {
            // (images ntile + 2 b: hi plane, ntile + 2 b + 1: lo plane of slot tile b, each behind its own [A] / [B] pair)
            float* mgs = red;                                              // [64 slots][16 rows]  (red: 9 x 256 floats)
            for (int b = 0; b < nslot_tiles; ++b) {
                bf16x8 pwp;                                                // marginals of slot tile b as the A operand (hi | lo)
    #pragma unroll
                for (int plane = 0; plane < 2; ++plane) {
                    const int t = ntile + 2 * b + plane;
                    lds_barrier();                                         // [A] pe image landed; red idle
                    if (t == ntile) {
                        if (wave < nslot_tiles) {
    #pragma unroll
                            for (int j = 0; j < 4; ++j) mgs[(16 * wave + r16) * 16 + 4 * kg + j] = (4 * kg + j < R) ? mgacc[j] : 0.f;
                        }
                        lds_barrier();                                     // [A'] table complete
                    }
                    if (plane == 0) {
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            uint16_t h, l;
                            split_bf16(mgs[(16 * b + ts0 + u) * 16 + r16], h, l);
                            pwp[u] = (short)h;
                            pwp[4 + u] = (short)l;
                        }
                    }
                    const char* img = ffbuf + (t & (NSLOT - 1)) * TILE_BYTES;
                    const unsigned img_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(img);
                    px_step(img_lds, pwp, acc);
                    lds_barrier();                                         // [B]
                }
            }
    
        } else {
            float* mgs = red;                                              // [64 slots][16 rows]  (red: 9 x 256 floats)
            for (int b = 0; b < nslot_tiles; ++b) {
                const int t = ntile + b;
                lds_barrier();                                             // [A] pe tile b landed; red idle
                if (b == 0) {
                    if (wave < nslot_tiles) {
    #pragma unroll
                        for (int j = 0; j < 4; ++j) mgs[(16 * wave + r16) * 16 + 4 * kg + j] = (4 * kg + j < R) ? mgacc[j] : 0.f;
                    }
                    lds_barrier();                                         // [A'] table complete
                }
                bf16x8 pwp;                                                // marginals of slot tile b as the A operand (hi | lo)
    #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint16_t h, l;
                    split_bf16(mgs[(16 * b + ts0 + u) * 16 + r16], h, l);
                    pwp[u] = (short)h;
                    pwp[4 + u] = (short)l;
                }
    #pragma unroll
                for (int plane = 0; plane < 2; ++plane) {
                    if (plane == 1) lds_barrier();                         // [B] hi-plane slot released
                    const char* img = (plane == 0 ? febuf : ffbuf) + (t & 1) * TILE_BYTES;
                    const unsigned img_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(img);
                    px_step(img_lds, pwp, acc);
                }
            }
    
        }
    }
    if (p.part_marg) {
        // ---- value-side pos-emb, round-6 form: the marginals leave the kernel (see RingParams::part_marg).  MG (fp32, accumulator layout
        // of the waves that own the slot blocks) is tabled in LDS as [compact slot][row], then every compute thread writes its share of
        // the [R][marg_slots] fp16 block of this workgroup in absolute slot order, normalised by the row's l.
        float* mgs = red;                                                  // [64 slots][16 rows]  (red: 9 x 256 floats)
        float* lrow = alpha_s;                                             // [16] l of the global rows (wave 0's copy)
        lds_barrier();                                                     // [M1] red idle (every wave past its last softmax)
        if (wave < nslot_tiles) {
#pragma unroll
            for (int j = 0; j < 4; ++j) mgs[(16 * wave + r16) * 16 + 4 * kg + j] = (4 * kg + j < R) ? mgacc[j] : 0.f;
        }
        if (wave == 0 && kg == 0) lrow[r16] = 1.0f / fmaxf(l_run, 1.0e-30f);
        lds_barrier();                                                     // [M2] table complete
        const int S = p.marg_slots, f0 = t1_first * p.kt;
        _Float16* dst = p.part_marg + (long)part * R * S;
        for (int e = ctid; e < R * S; e += 64 * kRingC) {
            const int row = e / S, sl = e - row * S;
            int cs = -1;                                                   // compact slot of absolute slot sl (none: the workgroup never touched it)
            if (sl < p.T) {
                const int f = sl - f0;
                if (f >= 0 && f < kMaxFramesPerWg) cs = f;
            } else if (sl < p.T + p.H) {
                const int v = ymap[sl - p.T];
                if (v != 255) cs = v;
            } else if (sl < p.T + p.H + p.W) {
                const int v = xmap[sl - p.T - p.H];
                if (v != 255) cs = v;
            }
            dst[e] = cs >= 0 ? (_Float16)(mgs[cs * 16 + row] * lrow[row]) : (_Float16)0.f;
        }
    }
    lds_barrier();                                                     // [E] every wave done with the ring
    HICOM_TR(0); HICOM_TR(1);   // tail: value-side pos-emb done
    // ---- partial global state of this workgroup --------------------------------------------------
    const long prow = (long)part * 16;
    if (wave == 0 && kg == 0 && r16 < R) {
        p.part_m[prow + r16] = m_run;
        p.part_l[prow + r16] = l_run;
    }
    // The accumulator rows go out through the (now idle) ring, regrouped so that every lane stores 16
    // contiguous bytes.
    {
        float* est = reinterpret_cast<float*>(smem) + wave * (R * SLICE);          // wave-private [R][SLICE]
        // fp16 form: the partial CONTEXT acc / l (the merge weighs it with l e^(m - M)): half the bytes of the 9 MB of partial states
        // this launch leaves dirty and the merge launch pulls back in.  1 / l of the accumulator rows 4 kg + j through the
        // wave-private LDS hop of the rescale factors.
        f32x4 linv = f32x4{1.f, 1.f, 1.f, 1.f};
        if (p.part_ctx16) {
            if (kg == 0) ascr[r16] = 1.0f / fmaxf(l_run, 1.0e-30f);
            linv = *reinterpret_cast<const f32x4*>(ascr + 4 * kg);
        }
        if (p.part_ctx16) {
            // fp16 rows through a wave-private LDS hop: 36 two-byte LDS writes in the accumulator layout, then 16-byte row chunks out --
            // 3 global store instructions per wave where the straight form had 36 (2-byte stores, 32 contiguous bytes each: 3.2k clocks
            // at the very end of the launch, tools/fused_trace.py; round 4's fp32 regroup: 4.3k).  Row pitch SLICE + 8 halves: 16-byte
            // aligned chunks, and the four accumulator rows of a lane group land on different banks.
            constexpr int RS = SLICE + 8;
            _Float16* e16 = reinterpret_cast<_Float16*>(smem) + wave * (12 * RS);        // wave-private [R <= 12][RS]
#pragma unroll
            for (int cb = 0; cb < KS; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * kg + j < R) e16[(4 * kg + j) * RS + 16 * cb + r16] = (_Float16)fminf(fmaxf(acc[cb][j] * linv[j], -65504.f), 65504.f);
            constexpr int C8 = SLICE / 8;                                                 // 16-byte chunks per row of this wave's slice
            static_assert(SLICE % 8 == 0, "16-byte chunks");
            const int n8 = R * C8;
            for (int it = lane; it < n8; it += 64) {
                const int row = it / C8, c8 = it - row * C8;
                *reinterpret_cast<u32x4*>(p.part_ctx16 + (prow + row) * E + ch_base + 8 * c8) = *reinterpret_cast<const u32x4*>(e16 + row * RS + 8 * c8);
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < KS; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * kg + j < R) est[(4 * kg + j) * SLICE + 16 * cb + r16] = acc[cb][j];
            const int n4 = R * (SLICE / 4);
            for (int it = lane; it < n4; it += 64) {
                const int row = it / (SLICE / 4), c4 = it - row * (SLICE / 4);
                *reinterpret_cast<f32x4*>(p.part_acc + (prow + row) * E + ch_base + 4 * c4) = *reinterpret_cast<const f32x4*>(est + 4 * it);
            }
        }
    }
    HICOM_TR(0); HICOM_TR(1);   // tail: state written (stores issued)
#ifdef HICOM_TRACE
    if (lane == 0 && wave == 0 && tr_n < 256) g_fused_trace[(blockIdx.x * 3 + 0) * 256 + tr_n] = tr_entry;
    if (lane == 0 && wave == 0) {
        g_fused_trace[(blockIdx.x * 3 + 0) * 256 + 250] = tr_entry_rt;
        g_fused_trace[(blockIdx.x * 3 + 0) * 256 + 251] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_fused_trace[(blockIdx.x * 3 + 0) * 256 + 252] = xcc & 15;
    }
#endif
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
    const int slots = fused_num_cus();                           // one resident workgroup per CU
    int wpw = (n_windows + slots - 1) / slots;                   // equal windows per workgroup
    // (HICOM_RING_WPW: dev -- more windows per workgroup = fewer streaming CUs: is the launch bound by the chip's HBM or by what one CU pulls?)
    static const int wpw_env = getenv("HICOM_RING_WPW") ? atoi(getenv("HICOM_RING_WPW")) : 0;
    if (wpw_env > wpw) wpw = wpw_env;
    if (wpw > kMaxWinPerWg) wpw = kMaxWinPerWg;
    return (n_windows + wpw - 1) / wpw;
}

static size_t ring_lds_bytes(int rows, int H, int W) {
    return (size_t)4 * 9 * 4096 + (size_t)(kRingC + 1) * 1024 + (size_t)kRingC * kRegroup * 4 + (size_t)kRingC * 64 +
           (64 + kMaxWinPerWg + 65 + 32 + 16) * 4 + (size_t)rows * (kMaxFramesPerWg + H + W) * 4;   // = the carve-out at the top of the kernel
}

extern "C" int hicom_fused_stream_fwd(const void* ff, const void* fe, const float* local_logits, int32_t T, int32_t H, int32_t W, int32_t E,
                                      int32_t kt, int32_t ks, const void* q_hi, const void* q_lo, int32_t rows,
                                      float l_scale, float l_bias, const float* pos_a, int32_t pos_stride,
                                      const void* pe_hi, const void* pe_lo,
                                      int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                      float* part_m, float* part_l,
                                      float* part_acc, int32_t nparts, float* ctx_local, void* ctx_hi,
                                      void* ctx_lo, void* ctx_f16, void* zero_ptr, int64_t zero_bytes, void* part_ctx_f16,
                                      void* part_marg_f16, int32_t marg_slots, void* stream) {
    HICOM_REQUIRE(ff && (fe || local_logits) && q_hi && q_lo && part_m && part_l && (part_acc || part_ctx_f16) && ((pe_hi && pe_lo) || part_marg_f16 || !pos_a) &&
                      (pos_a || !(pe_hi || part_marg_f16)), HICOM_EINVAL, "fused_stream: NULL pointer");
    HICOM_REQUIRE(!part_marg_f16 || (!pe_hi && !pe_lo && part_ctx_f16 && marg_slots >= T + H + W && marg_slots % 8 == 0 && (uintptr_t)part_marg_f16 % 16 == 0), HICOM_EINVAL,
                  "fused_stream: part_marg_f16 goes with part_ctx_f16, without pe planes, marg_slots >= T + H + W = %d (a multiple of 8)", T + H + W);
    HICOM_REQUIRE(ctx_local || (ctx_hi && ctx_lo) || ctx_f16, HICOM_EINVAL, "fused_stream: no local output");
    HICOM_REQUIRE(!ctx_hi == !ctx_lo, HICOM_EINVAL, "fused_stream: ctx_hi and ctx_lo go together");
    HICOM_REQUIRE(E == 1152, HICOM_EUNSUP, "fused_stream: E=%d (only 1152)", E);
    HICOM_REQUIRE(T > 0 && H > 0 && W > 0 && kt > 0 && ks > 0 && T % kt == 0 && H % ks == 0 && W % ks == 0, HICOM_EUNSUP,
                  "fused_stream: windows must partition the [%d,%d,%d] grid exactly", T, H, W);
    const int wsz = kt * ks * ks;
    HICOM_REQUIRE(wsz >= 16 && wsz <= 64, HICOM_EUNSUP, "fused_stream: window of %d tokens (16..64 supported)", wsz);
    HICOM_REQUIRE(rows > 0 && rows <= 12, HICOM_EUNSUP, "fused_stream: %d global rows (<= 12 supported)", rows);
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
    HICOM_REQUIRE(H <= 64 && W <= 64, HICOM_EUNSUP, "fused_stream: grid %dx%d exceeds the pos-emb slot maps", H, W);
    {   // compact slots a workgroup can touch: 8 frames + the rows and columns of its windows
        const int rows_t = ((wpw + (W / ks) - 2) / (W / ks) + 1) * ks, cols_t = wpw * ks;
        HICOM_REQUIRE(kMaxFramesPerWg + (rows_t < H ? rows_t : H) + (cols_t < W ? cols_t : W) <= 64, HICOM_EUNSUP,
                      "fused_stream: %d windows per workgroup touch too many pos-emb slots", wpw);
    }
    HICOM_REQUIRE(rows * (kMaxFramesPerWg + H + W) <= 64 * kRingC * kPosPerThread, HICOM_EUNSUP, "fused_stream: pos-emb table too large");
    const size_t smem = ring_lds_bytes(rows, H, W) + (local_logits ? (size_t)(kMaxWinPerWg + 64) * 4 : 0);   // (+ the token-index tables)
    HICOM_REQUIRE(smem <= 163840, HICOM_EUNSUP, "fused_stream: H + W = %d does not fit the LDS budget", H + W);
    RingParams p;
    p.ff = (const uint16_t*)ff; p.fe = (const uint16_t*)fe; p.llog = local_logits; p.T = T; p.H = H; p.W = W;
    p.kt = kt; p.ks = ks; p.nwy = H / ks; p.nwx = W / ks; p.NW = NW; p.WSZ = wsz;
    p.qhi = (const uint16_t*)q_hi; p.qlo = (const uint16_t*)q_lo; p.R = rows;
    p.l_scale = l_scale; p.l_bias = l_bias;
    p.pos_a = pos_a; p.pos_stride = pos_stride; p.t0i = t_index0; p.y0i = y_index0; p.x0i = x_index0;
    HICOM_REQUIRE(!part_ctx_f16 || (uintptr_t)part_ctx_f16 % 16 == 0, HICOM_EINVAL, "fused_stream: part_ctx_f16 alignment (16 bytes)");
    p.part_ctx16 = (_Float16*)part_ctx_f16;
    p.part_m = part_m; p.part_l = part_l; p.part_acc = part_acc; p.pe_hi = (const uint16_t*)pe_hi; p.pe_lo = (const uint16_t*)pe_lo;
    p.ctx_local = ctx_local; p.ctx_hi = (uint16_t*)ctx_hi; p.ctx_lo = (uint16_t*)ctx_lo; p.ctx_f16 = (_Float16*)ctx_f16; p.wpw = wpw;
    HICOM_REQUIRE(!zero_ptr || (zero_bytes > 0 && zero_bytes % 8 == 0 && (uintptr_t)zero_ptr % 8 == 0 && zero_bytes < (1 << 24)), HICOM_EINVAL,
                  "fused_stream: zero_ptr / zero_bytes");
    p.zero_ptr = (unsigned long long*)zero_ptr; p.zero_n = zero_ptr ? (int)(zero_bytes / 8) : 0;
    p.part_marg = (_Float16*)part_marg_f16; p.marg_slots = part_marg_f16 ? marg_slots : 0;
    static bool attr_set = false;
    if (!attr_set) {
        HICOM_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_ring_kernel<9, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 163840) == hipSuccess &&
                          hipFuncSetAttribute(reinterpret_cast<const void*>(fused_ring_kernel<9, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 163840) == hipSuccess,
                      HICOM_ELAUNCH, "fused_stream: 160 KiB of LDS per workgroup not available");
        attr_set = true;
    }
    // precomputed local logits win over frames_embed when both are given: frames_embed is then not read at all
    if (local_logits) HICOM_LAUNCH((fused_ring_kernel<9, true>), dim3((unsigned)nparts), dim3(kRingThreads), smem, (hipStream_t)stream,
                                   p.ff, p.fe, p.H, p.W, p.kt, p.ks, p.nwy, p.nwx, p.NW, p.WSZ, p.wpw, p);
    else HICOM_LAUNCH((fused_ring_kernel<9, false>), dim3((unsigned)nparts), dim3(kRingThreads), smem, (hipStream_t)stream,
                      p.ff, p.fe, p.H, p.W, p.kt, p.ks, p.nwy, p.nwx, p.NW, p.WSZ, p.wpw, p);
    return hicom_host::check_launch("fused_stream");
}
