// Native executor: ONE C call enqueues the whole compressor forward for a dense [T,H,W,E] input.
//
// The reference runs HIComProjector.forward (projector.py:676-708) as ~60 eager PyTorch ops on one stream.  Here the forward
// is a fixed plan of kernel launches over a caller-owned workspace, issued from C++ (no per-op Python / ctypes / allocator cost).
//
// Release recipe (use_guide = direct, exact window partition; `can_fuse`): FIVE launches on ONE stream, no events --
//
//   query_prep (q_proj -> granule hand-off -> fold: folded queries hi/lo, score-side pos table, local query rows, r0)
//   -> fused ring kernel (local windows + global attention: frames_embed and frames_feature each read once)
//   -> merge + v_proj (online-softmax merge of the workgroups' partial states, per-head v_proj as slab partials)
//   -> readout GEMM 1 (+ aux role: GELU(gc0 . o + r0), the global tail folded over out_proj)
//   -> readout GEMM 2 (+ aux role: last global readout layer -> the 32 global rows)             [+ newline rows]
//
// Other recipes of the plain family (guide off, window overlap, several query rows) take the two-kernel form: local windowed
// attention -> readout GEMMs on the main stream beside query prep -> MFMA token stream -> marginals / merge -> v_proj ->
// out_proj (+ residual) -> readout on a side stream, forked / joined through two caller-provided events (hipGraph-capturable,
// no state inside the library).  Recipes with adaptors (adapt_k / adapt_v: hicom_compressor_args.adapt) run the adaptor GEMMs
// in front of the local stage inside the same call.
// Phases can be run separately (HICOM_PHASE_*) so that the frame-sharded multi-GPU path can place its RCCL all-gather between
// them (STREAM on the main stream, FINISH on the comm stream; MERGE_ON_NEXT puts the merge on the comm stream as well).
#include <string.h>

#include <stdlib.h>
#include <utility>

#include "common.hpp"

using namespace hicom;

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct WsLayout {
    size_t ctx_local, hid_local, ctx_hi, ctx_lo, hid_hi, hid_lo, pooled_q, ctx16, hid16, ad_hid, ad_hid2, ad_ky, ad_vy, qp, qhi, qlo, pos_a, prep_state, tail_state, tail_sync, lq_inj, gq_inj, inj_l_s, inj_g_s, scores, part_m, part_l, part_acc, scratch, ml, acc,
        ctx_g, o, qres, pre, hid_g, tok, po, o_fix, r0, part_marg, total;
    int nw, R, rows_pad, nparts, P;
    long N, score_stride;
    bool marg;      // generic global path: the stream kernel keeps the positional marginals itself (no logit tensor; `scores` holds them)
};

// The fused local+global stream kernel applies to the release recipe: shared bf16 local query
// ("direct"), plain 1/sqrt(d) logits, windows that partition the grid, <= 14 folded global rows.
bool can_fuse(const hicom_compressor_args& a) {
    if (!(a.has_local && a.has_global) || !a.lq || a.lq_stride != 0 || a.lq_dt != HICOM_DT_BF16 || a.l2norm != 0 || a.gq_dt != HICOM_DT_BF16 ||
        a.inj_l.mode || a.inj_g.mode) return false;
    if (a.ak.w0 || a.av.w0) return false;         // adapted local streams: the two stages no longer share their tokens
    if (a.E != 1152 || a.nq * a.nh > 12) return false;
    for (const hicom_axis* x : {&a.at, &a.ay, &a.ax})
        if (x->n % x->k != 0 || x->nfull != x->nwin) return false;
    if (a.ay.k != a.ax.k) return false;
    const int wsz = a.at.k * a.ay.k * a.ax.k;
    if (wsz < 16 || wsz > 64 || a.H >= 256 || a.W >= 256) return false;
    const int R = a.nq * a.nh;
    const int nw = a.at.nwin * a.ay.nwin * a.ax.nwin, per_t = a.ay.nwin * a.ax.nwin;
    const int nparts = hicom_fused_stream_nparts(nw);
    const int wpw = (nw + nparts - 1) / nparts;
    // limits of fused_ring.hip: 160 KiB of LDS (ring 4 x 36 KiB + logit partials + tables), <= 32 windows
    // and <= 8 frames per workgroup, <= 1024 pos-emb table entries
    // (+ the token-index tables of the precomputed-logits variant, as hicom_fused_stream_fwd adds them)
    if (4 * 9 * 4096 + 9 * 1024 + 8 * 80 * 4 + 8 * 64 + (64 + 32 + 65 + 32 + 16) * 4 + R * (8 + a.H + a.W) * 4 +
            (a.local_logits ? (32 + 64) * 4 : 0) > 163840) return false;
    if (R * (8 + a.H + a.W) > 1024) return false;
    // compact pos-emb slots a workgroup can touch: 8 frames + the grid rows and columns of its windows (<= 64)
    const int rows_t = ((wpw + a.ax.nwin - 2) / a.ax.nwin + 1) * a.ay.k, cols_t = wpw * a.ax.k;
    if (a.H > 64 || a.W > 64 || 8 + (rows_t < a.H ? rows_t : a.H) + (cols_t < a.W ? cols_t : a.W) > 64) return false;
    return wpw <= 32 && ((wpw + per_t - 2) / per_t + 1) * a.at.k <= 8;
}

WsLayout make_layout(const hicom_compressor_args& a) {
    WsLayout w;
    memset(&w, 0, sizeof(w));
    w.nw = a.at.nwin * a.ay.nwin * a.ax.nwin;
    w.N = (long)a.T * a.H * a.W;
    w.R = a.nq * a.nh;
    w.rows_pad = (w.R + 15) / 16 * 16;
    w.P = a.P;
    w.score_stride = (w.N + 15) / 16 * 16;
    const bool fuse = a.has_global && can_fuse(a);
    w.nparts = !a.has_global ? 0 : (fuse ? hicom_fused_stream_nparts(w.nw) : hicom_global_stream_nparts(w.N, w.rows_pad));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    // zero-initialised-once region first (padding rows that no kernel ever writes)
    w.qhi = take((size_t)w.rows_pad * a.E * 2);
    w.qlo = take((size_t)w.rows_pad * a.E * 2);
    w.pos_a = take((size_t)w.rows_pad * (a.P > 0 ? a.P : 1) * 4);
    w.prep_state = take(a.has_global ? (size_t)hicom_query_prep_state_bytes(a.E) : 0);   // (epoch word + granules: zero once)
    w.tail_state = take(a.has_global ? (size_t)hicom_r16_chain_state_bytes(a.hidden) : 0);  // (the GEMV chain inside readout GEMM 2's launch: zero once)
    w.tail_sync = take(a.has_global && a.has_local ? (size_t)hicom_readout_tail_state_bytes() : 0);   // (counters of the fused tail launch: zero once)
    if (a.has_local) {
        w.ctx_local = take((size_t)w.nw * a.E * 4);          // fp32 form (two-kernel path) ...
        w.hid_local = take((size_t)w.nw * a.hidden * 4);
        w.ctx_hi = w.ctx_local;                               // ... or bf16 hi/lo planes in the same bytes (fused path)
        w.ctx_lo = w.ctx_local + (size_t)w.nw * a.E * 2;
        w.hid_hi = w.hid_local;
        w.hid_lo = w.hid_local + (size_t)w.nw * a.hidden * 2;
        w.pooled_q = take(a.lq ? 0 : (size_t)w.nw * a.E * 4);
        w.ctx16 = take((size_t)w.nw * a.E * 2);             // fp16 planes of the two-kernel path's readout (contexts, hidden layer)
        w.hid16 = take((size_t)w.nw * a.hidden * 2);
        // adaptor streams: the hidden layer of the MLP (shared by the two streams, which run one after the other) and y = MLP(x)
        // per adapted stream, fp16 [N, E]
        // (an adaptor whose y the CALLER supplies -- hicom_adaptor.y, the training forward -- needs none of this: 107 MB per region at 64
        // frames.  Whether y is supplied is part of what a plan is built for: the layout is the same on every call of one plan.)
        const bool mk = a.ak.w0 && !a.ak.y, mv = a.av.w0 && !a.av.y;
        if (mk || mv) w.ad_hid = take((size_t)w.N * a.E * 2);
        if (mk && mv) w.ad_hid2 = take((size_t)w.N * a.E * 2);      // (both adaptors: the two hidden layers live side by side, paired launches)
        if (mk) w.ad_ky = take((size_t)w.N * a.E * 2);
        if (mv) w.ad_vy = take((size_t)w.N * a.E * 2);
    }
    // in-call guide injection (hicom_injector): the injected rows and the injector's intermediates (coarse: two [1, <= 2E] rows; fine: the
    // projected queries, the attention output and out_proj's result [M, E] each, the projected text tokens [<= 64, E] twice)
    auto inj_scratch = [&](const hicom_compressor_args::hicom_injector& j, long M) -> size_t {
        if (j.mode == 1) return (size_t)(j.c_hidden + 2 * a.E) * 4 + 512;
        if (j.mode == 2) return ((size_t)3 * M + 2 * 64) * a.E * 4 + 5 * 256;
        return 0;
    };
    if (a.has_local && a.inj_l.mode) {
        w.lq_inj = take((size_t)w.nw * a.E * 4);
        w.inj_l_s = take(inj_scratch(a.inj_l, w.nw));
    }
    if (a.has_global && a.inj_g.mode) {
        w.gq_inj = take((size_t)a.nq * a.E * 4);
        w.inj_g_s = take(inj_scratch(a.inj_g, a.nq));
    }
    if (a.has_global) {
        w.qp = take((size_t)a.nq * a.E * 4);
        // (sized for the logit tensor also when the stream kernel keeps the positional marginals itself and writes its small
        // part_marg here instead: the layout must not depend on the HICOM_GLOBAL_NARROW test switch, which a cached plan may outlive)
        w.marg = !fuse && a.pe && hicom_global_stream_has_marg(w.N, a.E, w.rows_pad, a.H, a.W, w.nparts) == 1;
        {
            const size_t sb = (size_t)w.rows_pad * w.score_stride * 4;
            const size_t mb = a.pe && a.H > 0 && a.W > 0 ? (size_t)w.nparts * w.rows_pad * hicom_global_stream_marg_width(a.H, a.W) * 4 : 0;
            w.scores = take(sb > mb ? sb : mb);
        }
        w.part_m = take((size_t)w.nparts * w.rows_pad * 4);
        w.part_l = take((size_t)w.nparts * w.rows_pad * 4);
        w.part_acc = take((size_t)w.nparts * w.rows_pad * a.E * 4);
        w.scratch = take((size_t)w.R * a.T * (a.H + a.W + 2) * 4);
        w.ml = take((size_t)w.R * 2 * 4);
        w.acc = take((size_t)w.R * a.E * 4);
        w.ctx_g = take((size_t)w.R * a.E * 4);
        w.o = take((size_t)a.nq * a.E * 4);
        w.qres = take((size_t)a.nq * a.E * 4);
        w.pre = take((size_t)a.nq * a.E * 4);
        w.hid_g = take((size_t)a.nq * a.hidden * 4);
        w.tok = take((size_t)a.nq * a.hidden * 4);
        w.po = take((size_t)(a.E / 64) * a.E * 4);
        w.o_fix = take((size_t)a.E * 8);
        w.r0 = take((size_t)a.hidden * 4);
        // (release step, round 6: normalised positional marginals of the ring kernel's partial states, fp16 [nparts][R][marg_slots])
        w.part_marg = take(a.vpe_f16 && a.marg_slots > 0 ? (size_t)w.nparts * w.R * a.marg_slots * 2 : 0);
    }
    w.total = off;
    return w;
}

#ifdef HICOM_HOSTTIME   // dev-only: host microseconds per call site of the executor (tools/hosttime.py builds this variant)
#include <stdio.h>
#include <time.h>
static double g_ht[512], g_hgap[512];
static long g_hn[512];
static double g_hlast;
static inline double ht_now() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
struct HtDump {
    ~HtDump() {
        for (int i = 0; i < 512; ++i)
            if (g_hn[i]) fprintf(stderr, "[hosttime] executor.hip:%3d n=%6ld call %.2f us, since previous site %.2f us\n", i, g_hn[i], g_ht[i] / g_hn[i], g_hgap[i] / g_hn[i]);
    }
} g_htdump;
#define CHK(call)                                             \
    do {                                                      \
        const double t0_ = ht_now();                          \
        int rc_ = (call);                                     \
        const double t1_ = ht_now();                          \
        g_ht[__LINE__ % 512] += t1_ - t0_;                    \
        g_hgap[__LINE__ % 512] += t0_ - g_hlast;              \
        g_hn[__LINE__ % 512]++;                               \
        g_hlast = t1_;                                        \
        if (rc_ != HICOM_OK) return rc_;                      \
    } while (0)
#else
#define CHK(call)                  \
    do {                           \
        int rc_ = (call);          \
        if (rc_ != HICOM_OK) return rc_; \
    } while (0)
#endif

// HICOM_SHARD_TAIL=0: the frame-sharded step in round 4's form (merge on the comm stream, five small launches in FINISH): A/B switch
bool shard_tail_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HICOM_SHARD_TAIL");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// Frame-sharded release step in the four-launch form (round 5): the shard's state (M, L, ACC) comes out of the merge ROLE of GEMM 1's
// launch, r0 travels to the FINISH phase through r0_buf.  ONE predicate for the STREAM call and for hicom_compressor_takes_shard4 (the
// caller clears r0_buf in both blocks when it is false, so FINISH never consumes an r0 that STREAM did not write: ADVICE r5).
bool shard4_form(const hicom_compressor_args& a, const WsLayout& w) {
    const bool do_stream = a.phases & HICOM_PHASE_STREAM, do_finish = a.phases & HICOM_PHASE_FINISH;
    const bool fused = do_stream && can_fuse(a);
    const bool merge_on_next = fused && (a.phases & HICOM_PHASE_MERGE_ON_NEXT);
    const bool f16 = a.lw0_f16 && a.lw2_f16;
    const bool solo = a.state_out == nullptr;
    const bool prep1 = a.nq == 1 && a.E % 128 == 0 && a.E <= 1536 && a.E / a.nh <= 128;
    return fused && f16 && merge_on_next && !do_finish && !solo && prep1 && a.gc0 && a.hidden <= 1536 && a.hidden % 8 == 0 && a.r0_buf && a.local_out &&
           w.nparts <= 256 && a.E / a.nh <= 128 && a.E % 64 == 0 && shard_tail_enabled();
}

int check_args(const hicom_compressor_args& a) {
    HICOM_REQUIRE(a.has_local || a.has_global, HICOM_EINVAL, "compressor: nothing to do");
    HICOM_REQUIRE(a.ff && a.out && a.ws, HICOM_EINVAL, "compressor: NULL pointer");
    HICOM_REQUIRE(a.T > 0 && a.H > 0 && a.W > 0 && (a.E == 1152 || a.E == 768), HICOM_EINVAL, "compressor: bad input shape");
    HICOM_REQUIRE(a.hidden > 0 && a.hidden % 64 == 0, HICOM_EUNSUP, "compressor: hidden size %d must be a multiple of 64", a.hidden);
    if (a.has_local) HICOM_REQUIRE(a.lw0 && a.lw2, HICOM_EINVAL, "compressor: local readout weights");
    for (const auto* ad : {&a.ak, &a.av})
        if (ad->w0) {
            HICOM_REQUIRE(a.has_local && ad->w2_f16 && ad->gamma && ad->beta && ad->alpha, HICOM_EINVAL, "compressor: adaptor weights");
            HICOM_REQUIRE(a.l2norm == 0 && a.E % 64 == 0, HICOM_EUNSUP, "compressor: adaptors with clip-scale / E %% 64 != 0 run operator by operator");
        }
    if (a.has_global) {
        HICOM_REQUIRE((a.gq || a.inj_g.mode) && a.nq > 0 && a.nh > 0 && a.wq && a.wk && a.wv && a.wo && a.gw0 && a.gw2, HICOM_EINVAL,
                      "compressor: global weights");
        HICOM_REQUIRE(!a.inj_g.mode || a.inj_g.visual, HICOM_EINVAL, "compressor: the global injector needs the rows it injects into");
        HICOM_REQUIRE(a.n_global_rows >= a.nq && a.n_global_rows % a.nq == 0, HICOM_EINVAL, "compressor: global row count");
        HICOM_REQUIRE(a.gq_dt == HICOM_DT_BF16 || a.gq_dt == HICOM_DT_F32, HICOM_EINVAL, "compressor: gq_dt");
        HICOM_REQUIRE((a.pe == nullptr) == (a.kpe == nullptr), HICOM_EINVAL, "compressor: pe and kpe go together");
    }
    if ((a.has_local && a.inj_l.mode) || (a.has_global && a.inj_g.mode)) {
        HICOM_REQUIRE((a.phases & (HICOM_PHASE_STREAM | HICOM_PHASE_FINISH)) == (HICOM_PHASE_STREAM | HICOM_PHASE_FINISH) && !a.state_sets && !a.state_out,
                      HICOM_EUNSUP, "compressor: in-call guide injection runs STREAM and FINISH in one call (the injected rows live in its workspace)");
        HICOM_REQUIRE(!(a.has_local && a.inj_l.mode) || (!a.lq && a.l2norm == 0), HICOM_EINVAL, "compressor: the local injector takes the pooled queries (lq NULL, no clip-scale)");
        HICOM_REQUIRE((a.inj_l.mode >= 0 && a.inj_l.mode <= 2) && (a.inj_g.mode >= 0 && a.inj_g.mode <= 2), HICOM_EINVAL, "compressor: injector mode");
    }
    if (a.has_local && a.has_global && (a.phases & HICOM_PHASE_STREAM))
        HICOM_REQUIRE(a.stream_side && a.ev_fork && a.ev_join, HICOM_EINVAL, "compressor: side stream and events required");
    return HICOM_OK;
}

// GuideInjector.forward with a plain injector (reference projector.py:369-397) for M visual rows x (x_dt, [M, E]) -> out f32 [M, E]; the
// operator sequence of hicom_amd/injector.py: inject(), launch for launch (same kernels, same results).
int run_injector(const hicom_compressor_args::hicom_injector& j, const void* x, int x_dt, int M, int E, float* out, char* scratch, hipStream_t st) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    if (j.mode == 1) {
        HICOM_REQUIRE(j.guide && j.guide_rows == 1 && j.c_w0 && j.c_w2 && j.ln_w && j.ln_b && j.c_hidden > 0, HICOM_EINVAL, "compressor: coarse injector arguments");
        float* h = reinterpret_cast<float*>(scratch);
        float* cs = reinterpret_cast<float*>(scratch + al((size_t)j.c_hidden * 4));
        CHK(hicom_linear_fwd(j.guide, HICOM_DT_BF16, j.c_w0, HICOM_DT_BF16, j.c_b0, HICOM_DT_BF16, nullptr, 0, 1, j.c_hidden, E, 0, 0, HICOM_ACT_GELU, h, st));
        CHK(hicom_linear_fwd(h, HICOM_DT_F32, j.c_w2, HICOM_DT_BF16, j.c_b2, HICOM_DT_BF16, nullptr, 0, 1, 2 * E, j.c_hidden, 0, 0, HICOM_ACT_NONE, cs, st));
        return hicom_row_ln_fwd(x, x_dt, E, cs, 0, cs + E, 0, j.ln_w, j.ln_b, HICOM_DT_BF16, nullptr, 0, E, nullptr, 0, j.eps, out, HICOM_DT_F32, E, M, E, st);
    }
    HICOM_REQUIRE(j.mode == 2 && j.guide && j.guide_rows > 0 && j.guide_rows <= 64 && j.wq && j.wk && j.wv && j.wo && j.ln_w && j.ln_b && j.nheads > 0 &&
                      E % j.nheads == 0, HICOM_EINVAL, "compressor: fine injector arguments (<= 64 text tokens)");
    const int L = j.guide_rows;
    const size_t me = al((size_t)M * E * 4), le = al((size_t)64 * E * 4);
    float* qp = reinterpret_cast<float*>(scratch);
    float* ao = reinterpret_cast<float*>(scratch + me);
    float* o = reinterpret_cast<float*>(scratch + 2 * me);
    float* kp = reinterpret_cast<float*>(scratch + 3 * me);
    float* vp = reinterpret_cast<float*>(scratch + 3 * me + le);
    auto rows = [&](const void* xx, int dt, const void* wgt, const void* bias, float* y) -> int {    // injector.py: linear_rows()
        if (M <= 64 || dt != HICOM_DT_F32 || E % 64)
            return hicom_linear_fwd(xx, dt, wgt, HICOM_DT_BF16, bias, HICOM_DT_BF16, nullptr, 0, M, E, E, 0, 0, HICOM_ACT_NONE, y, st);
        return hicom_readout_gemm_fwd(reinterpret_cast<const float*>(xx), wgt, bias, HICOM_DT_BF16, M, E, E, HICOM_ACT_NONE, y, HICOM_DT_F32, E, 0, 0, st);
    };
    CHK(rows(x, x_dt, j.wq, j.bq, qp));
    CHK(hicom_linear_fwd(j.guide, HICOM_DT_BF16, j.wk, HICOM_DT_BF16, j.bk, HICOM_DT_BF16, nullptr, 0, L, E, E, 0, 0, HICOM_ACT_NONE, kp, st));
    CHK(hicom_linear_fwd(j.guide, HICOM_DT_BF16, j.wv, HICOM_DT_BF16, j.bv, HICOM_DT_BF16, nullptr, 0, L, E, E, 0, 0, HICOM_ACT_NONE, vp, st));
    CHK(hicom_small_mha_fwd(qp, kp, vp, M, L, j.nheads, E / j.nheads, ao, st));
    CHK(rows(ao, HICOM_DT_F32, j.wo, j.bo, o));
    return hicom_row_ln_fwd(x, x_dt, E, nullptr, 0, o, E, j.ln_w, j.ln_b, HICOM_DT_BF16, nullptr, 0, E, nullptr, 0, j.eps, out, HICOM_DT_F32, E, M, E, st);
}

}  // namespace

extern "C" int64_t hicom_compressor_workspace_bytes(const hicom_compressor_args* a) {
    if (!a) return HICOM_EINVAL;
    return (int64_t)make_layout(*a).total;
}

extern "C" int64_t hicom_compressor_zero_prefix_bytes(const hicom_compressor_args* a) {
    if (!a) return HICOM_EINVAL;
    const WsLayout w = make_layout(*a);
    // a FINISH-only block (frame-sharded step): everything -- its fixed-point accumulators o_fix are cleared by the chain launch BEHIND
    // each use (hicom_aux_gemv.x_fixed_clear), so they have to start at zero
    if (!(a->phases & HICOM_PHASE_STREAM)) return (int64_t)w.total;
    return (int64_t)(a->has_local ? w.ctx_local : w.qp);      // qhi | qlo | pos_a | prep_state | tail_state | tail_sync
}

extern "C" int hicom_compressor_takes_shard4(const hicom_compressor_args* a) {
    if (!a) return 0;
    return shard4_form(*a, make_layout(*a)) ? 1 : 0;
}

extern "C" int hicom_compressor_handoff_failures(const hicom_compressor_args* a, int32_t* out, void* stream) {
    HICOM_REQUIRE(a && out && a->ws, HICOM_EINVAL, "handoff_failures: NULL pointer");
    const WsLayout w = make_layout(*a);
    out[0] = out[1] = 0;
    if (!a->has_global) return HICOM_OK;
    const char* ws = (const char*)a->ws;
    unsigned v[2] = {0, 0};
    hipStream_t st = (hipStream_t)stream;
    // word 2 of each state block: the sticky count of expired hand-off spins (query_prep.hip, readout16.hip: gemv_chain_role)
    HICOM_REQUIRE(hipMemcpyAsync(&v[0], ws + w.prep_state + 8, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                      hipMemcpyAsync(&v[1], ws + w.tail_state + 8, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                      hipStreamSynchronize(st) == hipSuccess, HICOM_ELAUNCH, "handoff_failures: device read");
    out[0] = (int32_t)v[0];
    out[1] = (int32_t)v[1];
    return HICOM_OK;
}

extern "C" int hicom_compressor_is_fused(const hicom_compressor_args* a) {
    if (!a) return 0;
    return ((a->phases & HICOM_PHASE_STREAM) && can_fuse(*a)) ? 1 : 0;
}

extern "C" int64_t hicom_compressor_ctx16_offset(const hicom_compressor_args* a) {
    if (!a || !a->has_local) return HICOM_EINVAL;
    if (!(a->phases & HICOM_PHASE_STREAM) || !(a->lw0_f16 && a->lw2_f16) || a->E % 64 != 0 || a->hidden % 64 != 0) return HICOM_EUNSUP;
    const WsLayout w = make_layout(*a);
    return (int64_t)((a->has_global && can_fuse(*a)) ? w.ctx_hi : w.ctx16);
}

extern "C" int hicom_compressor_fwd(const hicom_compressor_args* ap);
extern "C" int hicom_compressor_fwd2(const hicom_compressor_args* first, const hicom_compressor_args* second) {
    if (int rc = hicom_compressor_fwd(first)) return rc;
    return hicom_compressor_fwd(second);
}

extern "C" int hicom_compressor_fwd(const hicom_compressor_args* ap) {
    HICOM_REQUIRE(ap, HICOM_EINVAL, "compressor: NULL args");
    const hicom_compressor_args& a = *ap;
    (void)hicom_host::take_stop_event();      // (nothing pending from a call that failed half-way)
#ifdef HICOM_HOSTTIME
    g_hlast = ht_now();
#endif
    CHK(check_args(a));
    const WsLayout w = make_layout(a);
    HICOM_REQUIRE(a.ws_bytes >= (int64_t)w.total, HICOM_EINVAL, "compressor: workspace too small (%lld < %zu)",
                  (long long)a.ws_bytes, w.total);
    char* ws = (char*)a.ws;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    hipStream_t sm = (hipStream_t)a.stream_main, ss = (hipStream_t)a.stream_side;
    const bool both = a.has_local && a.has_global;
    const bool do_stream = a.phases & HICOM_PHASE_STREAM, do_finish = a.phases & HICOM_PHASE_FINISH;
    const bool fused = do_stream && can_fuse(a);
    HICOM_REQUIRE(!a.local_logits || fused, HICOM_EINVAL, "compressor: local_logits is an input of the fused stream kernel only (release recipe)");
    const bool merge_on_next = fused && (a.phases & HICOM_PHASE_MERGE_ON_NEXT);
    // HICOM_PHASE_NEXT_IS_MAIN: what follows the STREAM phase runs on THIS call's main stream (a joined frame-sharded step: both phases on the
    // caller's stream) -- no event wait between the phases; stream_next is ignored (the caller's stream may be the null stream: handle 0)
    const bool next_same = (a.phases & HICOM_PHASE_NEXT_IS_MAIN) != 0;
    // event records folded into the launches they follow (release recipe); HICOM_FOLD_EVENTS=0 keeps separate records
    static const bool fold_env = !(getenv("HICOM_FOLD_EVENTS") && getenv("HICOM_FOLD_EVENTS")[0] == '0');
    // ... but never while the main stream is being captured into a hipGraph: the stop event of hipExtLaunchKernelGGL is
    // not a captured event-record node, so the fork / join edges of the side stream would be missing from the graph
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(sm, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    const bool fold_ev = fused && fold_env && !capturing;
    bool join_folded = false, done_folded = false;
    if (a.phases & HICOM_PHASE_MERGE_ON_NEXT)
        HICOM_REQUIRE(fused && a.ev_done && (a.stream_next || next_same), HICOM_EINVAL, "compressor: MERGE_ON_NEXT needs the release recipe, ev_done and stream_next");
    // the global chain runs on the side stream only when there is local work to overlap it with
    hipStream_t sg = (both && do_stream) ? ss : sm;
    const float qscale = a.has_global ? 1.0f / sqrtf((float)(a.E / a.nh)) : 0.f;
    const bool solo = a.state_out == nullptr;   // single shard: the merge normalises, no combine needed
    float* ml_out = solo ? F(w.ml) : (float*)a.state_out;
    float* acc_out = solo ? F(w.ctx_g) : (float*)a.state_out + 2 * w.R;

    auto fork = [&]() -> int {
        HICOM_REQUIRE(hipEventRecord((hipEvent_t)a.ev_fork, sm) == hipSuccess, HICOM_ELAUNCH, "compressor: event record");
        HICOM_REQUIRE(hipStreamWaitEvent(ss, (hipEvent_t)a.ev_fork, 0) == hipSuccess, HICOM_ELAUNCH, "compressor: stream wait");
        return HICOM_OK;
    };
    // the global stage's injected queries: the caller's rows, or -- in-call injection -- rows this call makes (f32, in the workspace)
    const void* gq = a.gq;
    int gq_dt = a.gq_dt;
    if (a.has_global && a.inj_g.mode) {
        gq = ws + w.gq_inj;
        gq_dt = HICOM_DT_F32;
    }
    auto query_prep = [&](hipStream_t st, bool with_local_rows) -> int {
        CHK(hicom_linear_fwd(gq, gq_dt, a.wq, HICOM_DT_BF16, a.bq, HICOM_DT_BF16, nullptr, 0, a.nq, a.E, a.E,
                             0, 0, HICOM_ACT_NONE, F(w.qp), st));
        return hicom_fold_query_split_fwd(F(w.qp), a.wk, a.kpe, a.nq, a.nh, a.E, a.P, qscale, ws + w.qhi, ws + w.qlo,
                                          F(w.pos_a), a.P, with_local_rows ? a.lq : nullptr, w.R, 16 - w.R, st);
    };
    auto merge = [&](hipStream_t st) -> int {
        if (w.marg)
            return hicom_global_merge_marg_fwd(F(w.part_m), F(w.part_l), F(w.part_acc), F(w.scores), w.nparts, w.R, w.rows_pad, a.E, w.N,
                                               a.H, a.W, a.pe, a.t_index0, a.y_index0, a.x_index0, F(w.scratch), ml_out, acc_out,
                                               solo ? 1 : 0, st);
        return hicom_global_merge_fwd(F(w.part_m), F(w.part_l), F(w.part_acc), w.nparts, w.R, w.rows_pad, a.E,
                                      F(w.scores), w.score_stride, w.N, a.H, a.W, a.pe, a.t_index0, a.y_index0,
                                      a.x_index0, F(w.scratch), ml_out, acc_out, solo ? 1 : 0, st);
    };
    const bool readout16 = a.lw0_f16 && a.lw2_f16 && a.E % 64 == 0 && a.hidden % 64 == 0;
    bool ctx16_ready = false;      // the window kernel wrote the fp16 plane itself
    auto local_readout = [&](hipStream_t st) -> int {
        if (readout16) {
            // the hot path's GEMM (one fp16 plane per activation, cached fp16 weights): 12 us per layer at 1296 rows against 26 for the
            // fp32-input form
            if (!ctx16_ready) CHK(hicom_to_f16_fwd(F(w.ctx_local), HICOM_DT_F32, ws + w.ctx16, (int64_t)w.nw * a.E, st));
            CHK(hicom_readout16_gemm_fwd(ws + w.ctx16, a.lw0_f16, a.lb0, HICOM_DT_BF16, w.nw, a.hidden, a.E, HICOM_ACT_GELU, ws + w.hid16, nullptr, 0, 0, 0,
                                         0, nullptr, st));
            CHK(hicom_readout16_gemm_fwd(ws + w.hid16, a.lw2_f16, a.lb2, HICOM_DT_BF16, w.nw, a.hidden, a.hidden, HICOM_ACT_NONE, nullptr,
                                         a.local_out ? a.local_out : a.out, a.out_dt, a.local_out ? a.hidden : a.ldo,
                                         a.local_out ? 0 : a.local_row0, a.local_out ? 0 : a.nl_group, nullptr, st));
            if (a.nl_count > 0 && !a.local_out)
                CHK(hicom_scatter_rows_fwd(a.newline, a.newline_dt, 1, a.hidden, a.out, a.out_dt, a.ldo, a.nl_first, a.nl_step,
                                           0, a.nl_count, st));
            return HICOM_OK;
        }
        CHK(hicom_readout_gemm_fwd(F(w.ctx_local), a.lw0, a.lb0, HICOM_DT_BF16, w.nw, a.hidden, a.E, HICOM_ACT_GELU,
                                   F(w.hid_local), HICOM_DT_F32, a.hidden, 0, 0, st));
        CHK(hicom_readout_gemm_fwd(F(w.hid_local), a.lw2, a.lb2, HICOM_DT_BF16, w.nw, a.hidden, a.hidden, HICOM_ACT_NONE,
                                   a.local_out ? a.local_out : a.out, a.out_dt, a.local_out ? a.hidden : a.ldo,
                                   a.local_out ? 0 : a.local_row0, a.local_out ? 0 : a.nl_group, st));
        if (a.nl_count > 0 && !a.local_out)
            CHK(hicom_scatter_rows_fwd(a.newline, a.newline_dt, 1, a.hidden, a.out, a.out_dt, a.ldo, a.nl_first, a.nl_step,
                                       0, a.nl_count, st));
        return HICOM_OK;
    };

    if (fused) {
        // ---- release recipe: ONE streaming kernel reads frames_embed and frames_feature once ------
        // main: q_proj, fold (+ guide -> local rows of the A operand), fused stream      | fork |
        // main: readout GEMMs            side: merge -> (finish)                 | join |
        // fp16 readout (lw0_f16 / lw2_f16 given): contexts / hidden travel as ONE fp16 plane (in the bytes of the hi plane)
        const bool f16 = a.lw0_f16 && a.lw2_f16;
        // single-stream tail (solo call, one query row per head): no side stream, no events.  Cross-stream event hops cost
        // 12-16 us each on this platform (fork after the ring kernel, join at the end: profiles/r02_a timeline), more than the
        // overlap they bought; instead the global chain's small GEMVs ride INSIDE the two readout GEMM launches:
        //   merge+v_proj | GEMM 1 (+ out_proj GEMV) | GEMM 2 (+ global readout 0 GEMV) | global readout 2 -> 32 rows
        const bool single = f16 && !merge_on_next && do_finish && solo && a.nq == 1 && !a.state_sets && w.nparts <= 256 &&
                            a.E / a.nh <= 128 && a.E <= 1536;
        // two-stream form: the stream kernel overwrites the partial states: the previous call's merge (side stream, possibly
        // still running when that call deferred its join) has to be done with them
        if (a.ev_merge && !merge_on_next && !single)
            HICOM_REQUIRE(hipStreamWaitEvent(sm, (hipEvent_t)a.ev_merge, 0) == hipSuccess, HICOM_ELAUNCH, "compressor: stream wait");
        // one-launch query prep (q_proj + fold + pos table + local rows [+ r0]) when the injected query is one bf16 row
        const bool prep1 = a.nq == 1 && a.E % 128 == 0 && a.E <= 1536 && a.E / a.nh <= 128;
        const bool tail5 = single && prep1 && a.gc0;       // the global tail folded over out_proj (gc0): one dependent stage fewer
        const bool ro2_aux = tail5 && a.hidden <= 1536;    // ... and its last layer inside GEMM 2's launch (aux GEMV: K <= 1536)
        // Frame-sharded release step in the four-launch form (round 5): the shard's state (M, L, ACC) comes out of the merge ROLE of
        // GEMM 1's launch (no merge launch on the comm stream, no v_proj here), r0 travels to the FINISH phase through r0_buf.
        const bool shard4 = shard4_form(a, w);
        // Round 5: FOUR launches.  The merge of the partial states is independent of the local readout, and the global tail behind it
        // is two dependent single-row layers: the merge rides as a ROLE on the CUs readout GEMM 1's tile grid leaves idle, the two layers
        // as a chain role (in-launch granule hand-off) under GEMM 2 -- the merge launch (5.2 us) is gone from the step.
        // HICOM_TAIL_LAUNCHES=5 keeps round 4's five-launch form (A/B switch).
        static int tail_env = -1;
        if (tail_env < 0) {
            const char* e = getenv("HICOM_TAIL_LAUNCHES");
            tail_env = (e && e[0] == '5') ? 5 : (e && e[0] == '3') ? 3 : 4;
        }
        // Round 6: the chain role takes hidden layers up to 4096 wide (the 7B model's 3584: readout16.hip, BIG form): the fifth launch of
        // that width (the last global layer, a 25.7-MB GEMV) then rides under GEMM 2 -- where GEMM 2 is long enough to hide a role that
        // pulls 750 KB per CU in seven round trips (~40 us): 64 frames at hidden 3584: 153.1 -> 145.2 us; at 32 frames (BASELINE
        // configs[3]: GEMM 2 ~25 us) the role becomes the launch's critical path and gives the 8 us back (95.7 against 95.9 us), so
        // that shape keeps round 5's form.  HICOM_CHAIN_WIDE=0 / =1: never / always (A/B switch).
        static const int chain_wide_env = getenv("HICOM_CHAIN_WIDE") ? (getenv("HICOM_CHAIN_WIDE")[0] == '0' ? 0 : 1) : -1;
        const bool chain_wide = chain_wide_env == 1 || (chain_wide_env < 0 && w.nw >= 1000);
        const bool chain_ok = tail5 && (a.hidden <= 1536 || (chain_wide && a.hidden <= 4096 && a.hidden % 8 == 0));
        const bool tail4 = single && tail5 && chain_ok && tail_env <= 4;
        // Round 6: the value-side pos-emb OUT of the ring kernel (marginals out, no pe tiles behind the token stream) and into the merge ROLE
        // of GEMM 1's launch as a product with the weight-only table v_proj . pe^T (merge_item.hpp) -- the round-4 / round-5 verdicts' lever.
        // Built, parity-green, bit-stable, MEASURED: the ring kernel gains 0.4-1.2 us, GEMM 1's launch loses 1.1-1.4 us (the
        // merge role becomes its critical path), the step 0.3-0.6 us -- so it is OPT-IN: HICOM_RING_MARG=1.
        static const bool ring_marg_env = getenv("HICOM_RING_MARG") && getenv("HICOM_RING_MARG")[0] == '1';
        const bool marg_out = single && tail5 && tail_env <= 4 && a.pe && a.vpe_f16 && a.marg_slots == 8 * (a.E / 64) &&
                              a.T + a.H + a.W <= a.marg_slots && ring_marg_env;
        if (prep1)
            CHK(hicom_query_prep_fwd(a.gq, a.lq, a.wq, a.bq, a.wk, a.kpe, a.nh, a.E, a.P, qscale, ws + w.qhi, ws + w.qlo, F(w.pos_a), a.P,
                                     w.R, (tail5 || shard4) ? a.gw0 : nullptr, a.gb0, a.bo, a.hidden, shard4 ? a.r0_buf : F(w.r0), ws + w.prep_state, sm));
        else CHK(query_prep(sm, true));
        if (fold_ev && !merge_on_next && !single) hicom_host::set_stop_event(a.ev_fork);      // "record ev_fork" rides on the launch
        CHK(hicom_fused_stream_fwd(a.ff, a.fe ? a.fe : a.ff, a.local_logits, a.T, a.H, a.W, a.E, a.at.k, a.ay.k, ws + w.qhi, ws + w.qlo,
                                   w.R, a.l_scale, a.l_bias, a.pe ? F(w.pos_a) : nullptr, a.P, (a.pe && !marg_out) ? a.pe_hi : nullptr, (a.pe && !marg_out) ? a.pe_lo : nullptr, a.t_index0, a.y_index0,
                                   a.x_index0, F(w.part_m), F(w.part_l), F(w.part_acc),
                                   w.nparts, nullptr, f16 ? nullptr : ws + w.ctx_hi, f16 ? nullptr : ws + w.ctx_lo, f16 ? ws + w.ctx_hi : nullptr,
                                   single ? ws + w.o_fix : nullptr, single ? (int64_t)a.E * 8 : 0, (single || shard4) ? ws + w.part_acc : nullptr,
                                   marg_out ? ws + w.part_marg : nullptr, marg_out ? a.marg_slots : 0, sm));
        if (shard4) {
            hicom_r16_role r1;
            memset(&r1, 0, sizeof(r1));
            r1.kind = HICOM_ROLE_MERGE_VPROJ;
            r1.part_m = F(w.part_m); r1.part_l = F(w.part_l); r1.part_acc = ws + w.part_acc; r1.part_dt = HICOM_DT_F16;
            r1.nparts = w.nparts; r1.rows = w.R; r1.rows_pad = w.rows_pad; r1.E = a.E;
            r1.out_ml = ml_out; r1.out_ctx = acc_out; r1.ctx_unnorm = 1;                 // the shard STATE, straight into the send buffer
            CHK(hicom_readout16_gemm_role_fwd(ws + w.ctx_hi, a.lw0_f16, a.lb0, HICOM_DT_BF16, w.nw, a.hidden, a.E, HICOM_ACT_GELU,
                                              ws + w.hid_hi, nullptr, 0, 0, 0, 0, &r1, sm));
            if (fold_ev && a.ev_done && !(a.nl_count > 0)) {
                hicom_host::set_stop_event(a.ev_done);        // last main-stream launch of the STREAM-only call
                done_folded = true;
            }
            CHK(hicom_readout16_gemm_fwd(ws + w.hid_hi, a.lw2_f16, a.lb2, HICOM_DT_BF16, w.nw, a.hidden, a.hidden, HICOM_ACT_NONE,
                                         nullptr, a.local_out, a.out_dt, a.hidden, 0, 0, nullptr, sm));
            if (a.ev_done && done_folded) {
                if (a.stream_next && !next_same)
                    HICOM_REQUIRE(hipStreamWaitEvent((hipStream_t)a.stream_next, (hipEvent_t)a.ev_done, 0) == hipSuccess, HICOM_ELAUNCH, "compressor: stream wait");
            } else if (a.ev_done) {
                HICOM_REQUIRE(hipEventRecord((hipEvent_t)a.ev_done, sm) == hipSuccess, HICOM_ELAUNCH, "compressor: event record");
                if (a.stream_next && !next_same)
                    HICOM_REQUIRE(hipStreamWaitEvent((hipStream_t)a.stream_next, (hipEvent_t)a.ev_done, 0) == hipSuccess, HICOM_ELAUNCH, "compressor: stream wait");
            }
            return HICOM_OK;
        }
        if (single && tail4) {
            const int64_t* ofx = (const int64_t*)(ws + w.o_fix);
            hicom_r16_role r1;
            memset(&r1, 0, sizeof(r1));
            r1.kind = HICOM_ROLE_MERGE_VPROJ;
            r1.part_m = F(w.part_m); r1.part_l = F(w.part_l); r1.part_acc = ws + w.part_acc; r1.part_dt = HICOM_DT_F16;
            r1.nparts = w.nparts; r1.rows = w.R; r1.rows_pad = w.rows_pad; r1.E = a.E; r1.w_v = a.wv; r1.o_fix = (int64_t*)(ws + w.o_fix);
            r1.out_ml = F(w.ml); r1.out_ctx = F(w.ctx_g);
            if (marg_out) { r1.part_marg = ws + w.part_marg; r1.vpe_f16 = a.vpe_f16; r1.marg_slots = a.marg_slots; }
            hicom_r16_role r2;
            memset(&r2, 0, sizeof(r2));
            r2.kind = HICOM_ROLE_GEMV_CHAIN;
            r2.gemv = hicom_aux_gemv{nullptr, 0, 0, a.bv, a.gc0, F(w.r0), nullptr, a.hidden, a.E, HICOM_ACT_GELU, F(w.hid_g),
                                     HICOM_DT_F32, HICOM_DT_F32, nullptr, 0, 0, 0, 0, ofx};
            r2.gemv2 = hicom_aux_gemv{nullptr, 0, 0, nullptr, a.gw2, a.gb2, nullptr, a.hidden, a.hidden, HICOM_ACT_NONE, nullptr,
                                      HICOM_DT_BF16, HICOM_DT_BF16, a.out, a.out_dt, a.n_global_rows, a.ldo, a.global_row0, nullptr};
            r2.chain_state = ws + w.tail_state;
            // HICOM_TAIL_LAUNCHES=3 (opt-in, measured SLOWER: 24.5 us against 12.0 + 11.6 and the gap between them): both GEMMs
            // and both roles in one grid, the hidden plane handed over inside the launch row block by row block (readout16.hip:
            // readout_tail_kernel).  The seam between the GEMMs costs what the kernel boundary cost, and the merge -> chain role, now one
            // uninterrupted run of dependent round trips, ends last.
            const hicom_r16_gemm g1{ws + w.ctx_hi, a.lw0_f16, a.lb0, HICOM_DT_BF16, (int32_t)w.nw, a.hidden, a.E, HICOM_ACT_GELU, ws + w.hid_hi, nullptr, 0, 0, 0, 0};
            const hicom_r16_gemm g2{ws + w.hid_hi, a.lw2_f16, a.lb2, HICOM_DT_BF16, (int32_t)w.nw, a.hidden, a.hidden, HICOM_ACT_NONE, nullptr,
                                    a.local_out ? a.local_out : a.out, a.out_dt, a.local_out ? (int64_t)a.hidden : a.ldo,
                                    a.local_out ? 0 : a.local_row0, a.local_out ? 0 : a.nl_group};
            int rc3 = tail_env == 3 ? hicom_readout_tail_fwd(&g1, &g2, &r1, &r2, ws + w.tail_sync, sm) : HICOM_EUNSUP;
            if (rc3 == HICOM_EUNSUP) {
                CHK(hicom_readout16_gemm_role_fwd(g1.a, g1.w, g1.b, g1.b_dt, g1.M, g1.N, g1.K, g1.act, g1.out_f16, nullptr, 0, 0, 0, 0, &r1, sm));
                // a deferred call's completion event rides on the step's last launch (a separate record costs the main stream a packet of
                // its own: forward_deferred measured 4 us behind forward at N = 1)
                if (fold_ev && a.defer_join && a.ev_join && !(a.nl_count > 0 && !a.local_out)) {
                    hicom_host::set_stop_event(a.ev_join);
                    join_folded = true;
                }
                CHK(hicom_readout16_gemm_role_fwd(g2.a, g2.w, g2.b, g2.b_dt, g2.M, g2.N, g2.K, g2.act, nullptr, g2.y, g2.y_dt, g2.ldy, g2.row0, g2.nl_group, &r2, sm));
            } else {
                CHK(rc3);
            }
        } else if (single && tail5 && !ro2_aux && tail_env <= 4) {
            // wide hidden layers (3584: the 7B model's width -- the chain role's second layer takes K <= 1536): the merge still rides under
            // GEMM 1 as a role, the first layer of the global tail (GELU(C o + r0), a plain GEMV role from the fixed-point sums) under
            // GEMM 2; the last layer is the launch at the end of this function.  One launch (4.7 us at 32 frames) less than the form below.
            hicom_r16_role r1;
            memset(&r1, 0, sizeof(r1));
            r1.kind = HICOM_ROLE_MERGE_VPROJ;
            r1.part_m = F(w.part_m); r1.part_l = F(w.part_l); r1.part_acc = ws + w.part_acc; r1.part_dt = HICOM_DT_F16;
            r1.nparts = w.nparts; r1.rows = w.R; r1.rows_pad = w.rows_pad; r1.E = a.E; r1.w_v = a.wv; r1.o_fix = (int64_t*)(ws + w.o_fix);
            r1.out_ml = F(w.ml); r1.out_ctx = F(w.ctx_g);
            if (marg_out) { r1.part_marg = ws + w.part_marg; r1.vpe_f16 = a.vpe_f16; r1.marg_slots = a.marg_slots; }
            CHK(hicom_readout16_gemm_role_fwd(ws + w.ctx_hi, a.lw0_f16, a.lb0, HICOM_DT_BF16, w.nw, a.hidden, a.E, HICOM_ACT_GELU,
                                              ws + w.hid_hi, nullptr, 0, 0, 0, 0, &r1, sm));
            hicom_aux_gemv ax{nullptr, 0, 0, a.bv, a.gc0, F(w.r0), nullptr, a.hidden, a.E, HICOM_ACT_GELU, F(w.hid_g),
                              HICOM_DT_F32, HICOM_DT_F32, nullptr, 0, 0, 0, 0, (const int64_t*)(ws + w.o_fix)};
            CHK(hicom_readout16_gemm_fwd(ws + w.hid_hi, a.lw2_f16, a.lb2, HICOM_DT_BF16, w.nw, a.hidden, a.hidden, HICOM_ACT_NONE,
                                         nullptr, a.local_out ? a.local_out : a.out, a.out_dt, a.local_out ? a.hidden : a.ldo,
                                         a.local_out ? 0 : a.local_row0, a.local_out ? 0 : a.nl_group, &ax, sm));
        } else if (single) {
            // merge + v_proj with the slab sums taken inside the launch (fixed-point accumulators, cleared by the stream kernel):
            // GEMM 1's aux role reads ONE 9-KB vector instead of E/64 partial vectors (83 KB per workgroup)
            // (the partial states travel as normalised fp16 contexts in the bytes of the fp32 accumulators: half of them)
            CHK(hicom_merge_vproj_fixed_fwd(F(w.part_m), F(w.part_l), ws + w.part_acc, HICOM_DT_F16, w.nparts, w.R, w.rows_pad, a.E, a.wv,
                                            (int64_t*)(ws + w.o_fix), F(w.ml), F(w.ctx_g), sm));
            // GEMM 1 carries the first dependent GEMV of the global tail: out_proj (+ residual), or -- tail5 -- out_proj and the
            // first readout layer as ONE layer, hid = GELU(gc0 . o + r0)
            const int64_t* ofx = (const int64_t*)(ws + w.o_fix);
            hicom_aux_gemv ax1{nullptr, 0, 0, a.bv, a.wo, a.bo, a.gq, a.E, a.E, HICOM_ACT_NONE, F(w.pre),
                               HICOM_DT_BF16, HICOM_DT_BF16, nullptr, 0, 0, 0, 0, ofx};
            if (tail5)
                ax1 = hicom_aux_gemv{nullptr, 0, 0, a.bv, a.gc0, F(w.r0), nullptr, a.hidden, a.E, HICOM_ACT_GELU, F(w.hid_g),
                                     HICOM_DT_F32, HICOM_DT_F32, nullptr, 0, 0, 0, 0, ofx};
            CHK(hicom_readout16_gemm_fwd(ws + w.ctx_hi, a.lw0_f16, a.lb0, HICOM_DT_BF16, w.nw, a.hidden, a.E, HICOM_ACT_GELU,
                                         ws + w.hid_hi, nullptr, 0, 0, 0, 0, &ax1, sm));
            // GEMM 2 carries the next one: the first readout layer, or -- tail5 -- the LAST one, written straight into the
            // 32 global rows of the output
            hicom_aux_gemv ax2{F(w.pre), 1, a.E, nullptr, a.gw0, a.gb0, nullptr, a.hidden, a.E, HICOM_ACT_GELU, F(w.hid_g),
                               HICOM_DT_BF16, HICOM_DT_BF16, nullptr, 0, 0, 0, 0, nullptr};
            if (ro2_aux)
                ax2 = hicom_aux_gemv{F(w.hid_g), 1, a.hidden, nullptr, a.gw2, a.gb2, nullptr, a.hidden, a.hidden, HICOM_ACT_NONE, nullptr,
                                     HICOM_DT_BF16, HICOM_DT_BF16, a.out, a.out_dt, a.n_global_rows, a.ldo, a.global_row0, nullptr};
            CHK(hicom_readout16_gemm_fwd(ws + w.hid_hi, a.lw2_f16, a.lb2, HICOM_DT_BF16, w.nw, a.hidden, a.hidden, HICOM_ACT_NONE,
                                         nullptr, a.local_out ? a.local_out : a.out, a.out_dt, a.local_out ? a.hidden : a.ldo,
                                         a.local_out ? 0 : a.local_row0, a.local_out ? 0 : a.nl_group, (tail5 && !ro2_aux) ? nullptr : &ax2, sm));
        }
        if (single) {
            if (a.nl_count > 0 && !a.local_out)
                CHK(hicom_scatter_rows_fwd(a.newline, a.newline_dt, 1, a.hidden, a.out, a.out_dt, a.ldo, a.nl_first, a.nl_step,
                                           0, a.nl_count, sm));
            if (!ro2_aux && !tail4)
                CHK(hicom_linear_to_rows_fwd(F(w.hid_g), HICOM_DT_F32, a.gw2, HICOM_DT_BF16, a.gb2, HICOM_DT_BF16, a.nq, a.hidden,
                                             a.hidden, HICOM_ACT_NONE, a.out, a.out_dt, a.ldo, a.global_row0, a.n_global_rows, sm));
            if (a.defer_join && a.ev_join && !join_folded)      // (a deferred call's completion event: everything is on the main stream here)
                HICOM_REQUIRE(hipEventRecord((hipEvent_t)a.ev_join, sm) == hipSuccess, HICOM_ELAUNCH, "compressor: event record");
            if (a.ev_done) {
                HICOM_REQUIRE(hipEventRecord((hipEvent_t)a.ev_done, sm) == hipSuccess, HICOM_ELAUNCH, "compressor: event record");
                if (a.stream_next && !next_same)
                    HICOM_REQUIRE(hipStreamWaitEvent((hipStream_t)a.stream_next, (hipEvent_t)a.ev_done, 0) == hipSuccess, HICOM_ELAUNCH,
                                  "compressor: stream wait");
            }
            return HICOM_OK;
        }
        if (!merge_on_next) {
            if (fold_ev) HICOM_REQUIRE(hipStreamWaitEvent(ss, (hipEvent_t)a.ev_fork, 0) == hipSuccess, HICOM_ELAUNCH, "compressor: stream wait");
            else CHK(fork());
            // the value-side pos-emb is already inside the partial contexts: a plain merge, one launch
            if (fold_ev && a.ev_merge) hicom_host::set_stop_event(a.ev_merge);
            CHK(hicom_global_merge_fwd(F(w.part_m), F(w.part_l), F(w.part_acc), w.nparts, w.R, w.rows_pad, a.E, nullptr, 0, w.N,
                                       a.H, a.W, nullptr, 0, 0, 0, nullptr, ml_out, acc_out, solo ? 1 : 0, ss));
            if (a.ev_merge && !fold_ev)
                HICOM_REQUIRE(hipEventRecord((hipEvent_t)a.ev_merge, ss) == hipSuccess, HICOM_ELAUNCH, "compressor: event record");
        }
        // readout MLP on bf16 planes: contexts (hi/lo) -> hidden (hi/lo) -> packed output rows
        if (f16) CHK(hicom_readout16_gemm_fwd(ws + w.ctx_hi, a.lw0_f16, a.lb0, HICOM_DT_BF16, w.nw, a.hidden, a.E, HICOM_ACT_GELU,
                                              ws + w.hid_hi, nullptr, 0, 0, 0, 0, nullptr, sm));
        else CHK(hicom_planes_gemm_fwd(ws + w.ctx_hi, ws + w.ctx_lo, a.lw0, a.lb0, HICOM_DT_BF16, w.nw, a.hidden, a.E,
                                       HICOM_ACT_GELU, ws + w.hid_hi, ws + w.hid_lo, nullptr, 0, 0, 0, 0, sm));
        if (fold_ev && a.ev_done && merge_on_next && !do_finish && !(a.nl_count > 0 && !a.local_out)) {
            hicom_host::set_stop_event(a.ev_done);        // last main-stream launch of a STREAM-only call
            done_folded = true;
        }
        if (f16) CHK(hicom_readout16_gemm_fwd(ws + w.hid_hi, a.lw2_f16, a.lb2, HICOM_DT_BF16, w.nw, a.hidden, a.hidden, HICOM_ACT_NONE,
                                              nullptr, a.local_out ? a.local_out : a.out, a.out_dt, a.local_out ? a.hidden : a.ldo,
                                              a.local_out ? 0 : a.local_row0, a.local_out ? 0 : a.nl_group, nullptr, sm));
        else CHK(hicom_planes_gemm_fwd(ws + w.hid_hi, ws + w.hid_lo, a.lw2, a.lb2, HICOM_DT_BF16, w.nw, a.hidden, a.hidden,
                                       HICOM_ACT_NONE, nullptr, nullptr, a.local_out ? a.local_out : a.out, a.out_dt,
                                       a.local_out ? a.hidden : a.ldo, a.local_out ? 0 : a.local_row0,
                                       a.local_out ? 0 : a.nl_group, sm));
        if (a.nl_count > 0 && !a.local_out)
            CHK(hicom_scatter_rows_fwd(a.newline, a.newline_dt, 1, a.hidden, a.out, a.out_dt, a.ldo, a.nl_first, a.nl_step,
                                       0, a.nl_count, sm));
    } else if (do_stream) {
        if (both) CHK(fork());
        auto global_prep = [&]() -> int {
            if (a.inj_g.mode)      // coarse / fine: the guide into the learnable queries (projector.py:642 with :369-397), 3 / 6 small launches
                CHK(run_injector(a.inj_g, a.inj_g.visual, HICOM_DT_BF16, a.nq, a.E, F(w.gq_inj), ws + w.inj_g_s, sg));
            if (!a.reuse_queries || a.inj_g.mode) CHK(query_prep(sg, false));        // (guide off: weight-only, kept in the workspace across calls)
            return HICOM_OK;
        };
        auto global_stream = [&]() -> int {
            if (w.marg)
                CHK(hicom_global_stream_marg_fwd(a.ff, w.N, a.E, ws + w.qhi, ws + w.qlo, w.R, w.rows_pad, F(w.pos_a), a.P, a.H, a.W,
                                                 a.t_index0, a.y_index0, a.x_index0, nullptr, 0, F(w.part_m), F(w.part_l), F(w.part_acc),
                                                 F(w.scores), w.nparts, sg));
            else
                CHK(hicom_global_stream_fwd(a.ff, w.N, a.E, ws + w.qhi, ws + w.qlo, w.R, w.rows_pad,
                                            a.pe ? F(w.pos_a) : nullptr, a.P, a.H, a.W, a.t_index0, a.y_index0, a.x_index0,
                                            F(w.scores), w.score_stride, F(w.part_m), F(w.part_l), F(w.part_acc), w.nparts, sg));
            return HICOM_OK;
        };
        auto global_stream_part = [&]() -> int {
            CHK(global_prep());
            return global_stream();
        };
        // Many query rows (guide off: 288): the stream kernel fills the chip for ~230 us and IS the critical path; run beside it, the
        // local window kernel takes CU time from it (276 against 233 us) while the latency-bound tail behind it (merge + four 32-row
        // linears, ~80 us) leaves the chip idle.  So: stream kernel first and alone, then the local chain beside the global tail.
        static int sf_env = -1;
        if (sf_env < 0) {
            const char* e = getenv("HICOM_STREAM_FIRST");            // dev / A-B switch: 0 = local and global chains side by side from the start
            sf_env = (e && e[0] == '0') ? 0 : 1;
        }
        const bool stream_first = sf_env && both && w.rows_pad > 16 && !(a.ak.w0 || a.av.w0) && a.ev_merge && sg == ss;
        const bool pool_q = a.has_local && !a.lq;                    // guide off: per-window pooled query
        if (stream_first) {
            CHK(global_stream_part());
            HICOM_REQUIRE(hipEventRecord((hipEvent_t)a.ev_merge, ss) == hipSuccess, HICOM_ELAUNCH, "compressor: event record");
            // (the 8-us pooling kernel in front of the wait: it runs under the stream kernel's start)
            if (pool_q) CHK(hicom_trilinear_pool_fwd(a.ff, a.T, a.H, a.W, a.E, a.at.nwin, a.ay.nwin, a.ax.nwin, F(w.pooled_q), sm));
            // (... and the local stage's guide injection is enqueued here too: 3 / 6 launches.  Beside the stream kernel, which holds every
            // CU, they mostly run when it ends -- tools/recipe_trace.py -- but ordering the stream kernel BEHIND them was measured no faster:
            // "fine" 0.474 against 0.463 ms)
            if (pool_q && a.inj_l.mode) CHK(run_injector(a.inj_l, F(w.pooled_q), HICOM_DT_F32, (int)w.nw, a.E, F(w.lq_inj), ws + w.inj_l_s, sm));
            HICOM_REQUIRE(hipStreamWaitEvent(sm, (hipEvent_t)a.ev_merge, 0) == hipSuccess, HICOM_ELAUNCH, "compressor: stream wait");
        }
        // Host enqueue order matters (each launch costs a few us of host time): the long local
        // attention kernel goes first so that the side chain is enqueued while it runs.
        if (a.has_local && a.ev_queries)      // (lq made by the caller on a stream of its own, beside the global stream kernel)
            HICOM_REQUIRE(hipStreamWaitEvent(sm, (hipEvent_t)a.ev_queries, 0) == hipSuccess, HICOM_ELAUNCH, "compressor: stream wait");
        if (a.has_local) {
            const void* q = a.lq;
            int q_dt = a.lq_dt;
            int64_t q_stride = a.lq_stride;
            if (!q) {   // guide off: per-window pooled query
                if (!stream_first) CHK(hicom_trilinear_pool_fwd(a.ff, a.T, a.H, a.W, a.E, a.at.nwin, a.ay.nwin, a.ax.nwin, F(w.pooled_q), sm));
                q = F(w.pooled_q);
                q_dt = HICOM_DT_F32;
                q_stride = a.E;
                if (a.inj_l.mode) {     // coarse / fine injection into the pooled queries (projector.py:542)
                    if (!stream_first) CHK(run_injector(a.inj_l, F(w.pooled_q), HICOM_DT_F32, (int)w.nw, a.E, F(w.lq_inj), ws + w.inj_l_s, sm));
                    q = F(w.lq_inj);
                }
            }
            if (a.ak.w0 || a.av.w0) {
                // k / v adaptors (projector.py:533-534): y = MLP(x) over all tokens on the dense MFMA GEMM (raw bf16 tokens x bf16
                // weights -> fp16 hidden with GELU -> fp16 y), then the window attention with LayerNorm + alpha blend fused into
                // its row loads: the blended streams are never written
                const void* key_x = a.fe ? a.fe : a.ff;
                auto mlp = [&](const hicom_compressor_args::hicom_adaptor& ad, const void* x, size_t y_off) -> int {
                    CHK(hicom_dense16_gemm_fwd(x, a.E, ad.w0, a.E, HICOM_DT_BF16, ad.b0, HICOM_DT_BF16, (int32_t)w.N, a.E, a.E, HICOM_ACT_GELU,
                                               ws + w.ad_hid, a.E, a.E, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, 0, 0, nullptr, 0,
                                               nullptr, sm));
                    return hicom_dense16_gemm_fwd(ws + w.ad_hid, a.E, ad.w2_f16, a.E, HICOM_DT_F16, ad.b2, HICOM_DT_BF16, (int32_t)w.N, a.E, a.E,
                                                  HICOM_ACT_NONE, ws + y_off, a.E, a.E, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, 0, 0,
                                                  nullptr, 0, nullptr, sm);
                };
                // both adaptors from scratch: each layer of the two MLPs as ONE paired launch (two problems of one shape: their tiles share the
                // last, partly filled round of workgroup slots -- 6 570 tiles = 6.4 rounds instead of 2 x 3.2)
                static const bool pair_env = !(getenv("HICOM_ADAPT_PAIR") && getenv("HICOM_ADAPT_PAIR")[0] == '0');
                if (a.ak.w0 && a.av.w0 && !a.ak.y && !a.av.y && pair_env && a.ak.b0 && a.av.b0 && a.ak.b2 && a.av.b2) {
                    CHK(hicom_dense16_gemm_pair_fwd(key_x, a.ak.w0, a.ak.b0, ws + w.ad_hid, nullptr, a.ff, a.av.w0, a.av.b0, ws + w.ad_hid2, nullptr,
                                                    a.E, a.E, HICOM_DT_BF16, HICOM_DT_BF16, (int32_t)w.N, a.E, a.E, HICOM_ACT_GELU, a.E, a.E, 0, sm));
                    CHK(hicom_dense16_gemm_pair_fwd(ws + w.ad_hid, a.ak.w2_f16, a.ak.b2, ws + w.ad_ky, nullptr, ws + w.ad_hid2, a.av.w2_f16, a.av.b2,
                                                    ws + w.ad_vy, nullptr, a.E, a.E, HICOM_DT_F16, HICOM_DT_BF16, (int32_t)w.N, a.E, a.E, HICOM_ACT_NONE,
                                                    a.E, a.E, 0, sm));
                } else {
                    if (a.ak.w0 && !a.ak.y) CHK(mlp(a.ak, key_x, w.ad_ky));
                    if (a.av.w0 && !a.av.y) CHK(mlp(a.av, a.ff, w.ad_vy));
                }
                const void* ky = !a.ak.w0 ? nullptr : (a.ak.y ? a.ak.y : (const void*)(ws + w.ad_ky));      // (.y: the caller's own MLP outputs)
                const void* vy = !a.av.w0 ? nullptr : (a.av.y ? a.av.y : (const void*)(ws + w.ad_vy));
                CHK(hicom_local_attn_adapt_fwd(key_x, ky, a.ak.gamma, a.ak.beta, a.ak.alpha,
                                               a.ff, vy, a.av.gamma, a.av.beta, a.av.alpha,
                                               a.adapt_alpha_dt, a.adapt_eps, a.E, a.at, a.ay, a.ax, q, q_dt, q_stride, a.l_scale, a.l_bias,
                                               F(w.ctx_local), sm));
            } else {
                // (fp16 readout: the contexts leave the window kernel as the fp16 plane the GEMM reads -- no conversion launch)
                ctx16_ready = readout16;
                CHK(hicom_local_attn_fwd(a.fe ? a.fe : a.ff, HICOM_DT_BF16, a.ff, HICOM_DT_BF16, a.E, a.at, a.ay, a.ax, q, q_dt, q_stride, a.l_scale,
                                         a.l_bias, a.l2norm, readout16 ? nullptr : F(w.ctx_local), readout16 ? ws + w.ctx16 : nullptr, sm));
            }
        }
        if (a.has_global) {
            if (!stream_first) CHK(global_stream_part());
            CHK(merge(sg));
        }
        if (a.has_local) CHK(local_readout(sm));
    }

    // ------------------------------------------------------------------ global chain, part 2
    if (a.ag_fn && do_finish && !do_stream) {
        // the all-gather of the exchange buffers, enqueued by this call (RCCL's ncclAllGather through the caller's communicator):
        // [world][ag_bytes] <- every rank's [ag_bytes]; everything below reads the gathered buffer in stream order behind it
        typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
        HICOM_REQUIRE(a.ag_comm && a.ag_send && a.ag_recv && a.ag_bytes > 0, HICOM_EINVAL, "compressor: all-gather arguments");
        typedef int (*group_fn)(void);
        const bool two = a.ag_bytes2 > 0;
        if (two) {
            HICOM_REQUIRE(a.ag_group_start && a.ag_group_end && a.ag_send2 && a.ag_recv2, HICOM_EINVAL, "compressor: grouped all-gather arguments");
            const int grc = ((group_fn)a.ag_group_start)();
            HICOM_REQUIRE(grc == 0, HICOM_ELAUNCH, "compressor: ncclGroupStart returned %d", grc);
        }
        const int nrc = ((allgather_fn)a.ag_fn)(a.ag_send, a.ag_recv, (size_t)a.ag_bytes, /* ncclUint8 */ 1, a.ag_comm, sm);
        int nrc2 = 0, nrc3 = 0;
        if (two) {
            nrc2 = ((allgather_fn)a.ag_fn)(a.ag_send2, a.ag_recv2, (size_t)a.ag_bytes2, 1, a.ag_comm, sm);
            nrc3 = ((group_fn)a.ag_group_end)();                     // (always closed: an open group would swallow every later collective)
        }
        HICOM_REQUIRE(nrc == 0 && nrc2 == 0 && nrc3 == 0, HICOM_ELAUNCH, "compressor: ncclAllGather / ncclGroupEnd returned %d / %d / %d", nrc, nrc2, nrc3);
    }
    const bool finish4 = a.has_global && do_finish && !do_stream && a.state_sets && a.nsets > 0 && a.nsets <= 256 && a.r0_buf && a.gc0 && a.nq == 1 &&
                         a.hidden <= 1536 && a.hidden % 8 == 0 && a.E % 64 == 0 && a.E / a.nh <= 128 && a.E <= 1536 && a.lw0_f16 && shard_tail_enabled();
    if (finish4) {
        // FINISH of the four-launch sharded step: [clear the accumulators | merge of the gathered shard states + v_proj | chain launch:
        // GELU(C o + r0) -> granule hand-off -> last readout layer -> the 32 global rows] -- three launches where the generic form
        // below has five (combine, v_proj, out_proj, two readout layers)
        HICOM_REQUIRE(a.state_set_stride >= (int64_t)(2 * w.R + (long)w.R * a.E), HICOM_EINVAL, "compressor: state set layout");
        // (no memset: the chain launch below clears the accumulators behind its own read of them -- x_fixed_clear -- and the workspace
        // starts zeroed: hicom_compressor_zero_prefix_bytes covers o_fix for a FINISH-only block)
        CHK(hicom_merge_vproj_sets_fwd((const float*)a.state_sets, a.state_set_stride, a.nsets, w.R, a.E, a.wv, (int64_t*)(ws + w.o_fix), nullptr, nullptr, sg));
        hicom_r16_role r2;
        memset(&r2, 0, sizeof(r2));
        r2.kind = HICOM_ROLE_GEMV_CHAIN;
        r2.gemv = hicom_aux_gemv{nullptr, 0, 0, a.bv, a.gc0, a.r0_buf, nullptr, a.hidden, a.E, HICOM_ACT_GELU, nullptr,
                                 HICOM_DT_F32, HICOM_DT_F32, nullptr, 0, 0, 0, 0, (const int64_t*)(ws + w.o_fix), 1};
        r2.gemv2 = hicom_aux_gemv{nullptr, 0, 0, nullptr, a.gw2, a.gb2, nullptr, a.hidden, a.hidden, HICOM_ACT_NONE, nullptr,
                                  HICOM_DT_BF16, HICOM_DT_BF16, a.out, a.out_dt, a.n_global_rows, a.ldo, a.global_row0, nullptr};
        r2.chain_state = ws + w.tail_state;
        CHK(hicom_gemv_chain_fwd(&r2, sg));
    } else if (a.has_global && do_finish) {
        const float* ctx = F(w.ctx_g);
        if (a.state_sets) {   // gathered (M, L, ACC) states of all shards: [nsets][2R + R*E]
            HICOM_REQUIRE(a.nsets > 0 && a.state_set_stride >= (int64_t)(2 * w.R + (long)w.R * a.E), HICOM_EINVAL,
                          "compressor: state set layout");
            CHK(hicom_global_combine_strided_fwd((const float*)a.state_sets, (const float*)a.state_sets + 2 * w.R,
                                                 a.state_set_stride, a.nsets, w.R, a.E, F(w.ctx_g), sg));
        }
        CHK(hicom_linear_fwd(ctx, HICOM_DT_F32, a.wv, HICOM_DT_BF16, a.bv, HICOM_DT_BF16, nullptr, 0, a.nq, a.E, a.E,
                             a.nh, a.E / a.nh, HICOM_ACT_NONE, F(w.o), sg));
        // residual with the injected query (projector.py:646), read in its own dtype
        CHK(hicom_linear_fwd(F(w.o), HICOM_DT_F32, a.wo, HICOM_DT_BF16, a.bo, HICOM_DT_BF16, gq, gq_dt == HICOM_DT_F32 ? 0 : 2, a.nq, a.E, a.E,
                             0, 0, HICOM_ACT_NONE, F(w.pre), sg));
        CHK(hicom_linear_fwd(F(w.pre), HICOM_DT_F32, a.gw0, HICOM_DT_BF16, a.gb0, HICOM_DT_BF16, nullptr, 0, a.nq, a.hidden,
                             a.E, 0, 0, HICOM_ACT_GELU, F(w.hid_g), sg));
        join_folded = fold_ev && both && do_stream && !merge_on_next && sg == ss;
        if (join_folded) hicom_host::set_stop_event(a.ev_join);                    // "record ev_join" rides on the last launch
        CHK(hicom_linear_to_rows_fwd(F(w.hid_g), HICOM_DT_F32, a.gw2, HICOM_DT_BF16, a.gb2, HICOM_DT_BF16, a.nq, a.hidden,
                                     a.hidden, HICOM_ACT_NONE, a.out, a.out_dt, a.ldo, a.global_row0, a.n_global_rows, sg));
    }

    if (both && do_stream && !merge_on_next) {
        if (!join_folded)
            HICOM_REQUIRE(hipEventRecord((hipEvent_t)a.ev_join, ss) == hipSuccess, HICOM_ELAUNCH, "compressor: event record");
        const bool defer = fused && a.defer_join && a.ev_merge;      // the caller joins on ev_join itself
        if (!defer)
            HICOM_REQUIRE(hipStreamWaitEvent(sm, (hipEvent_t)a.ev_join, 0) == hipSuccess, HICOM_ELAUNCH, "compressor: stream wait");
    }
    const bool fold_ev2 = fold_env && !capturing;
    if (a.place_src && do_finish) {
        const int esz = a.out_dt == HICOM_DT_F32 ? 4 : 2;
        if (fold_ev2 && a.ev_done && !do_stream) {
            hicom_host::set_stop_event(a.ev_done);
            done_folded = true;
        }
        CHK(hicom_place_blocks_fwd(a.place_src, a.place_block_rows, a.place_nblocks, a.place_block_stride, a.hidden * esz, a.out,
                                   a.ldo * esz, 0, a.nl_group, sm));
    }
    if (a.ev_done && done_folded) {
        if (a.stream_next && !next_same)
            HICOM_REQUIRE(hipStreamWaitEvent((hipStream_t)a.stream_next, (hipEvent_t)a.ev_done, 0) == hipSuccess, HICOM_ELAUNCH,
                          "compressor: stream wait");
    } else if (a.ev_done) {
        HICOM_REQUIRE(hipEventRecord((hipEvent_t)a.ev_done, sm) == hipSuccess, HICOM_ELAUNCH, "compressor: event record");
        if (a.stream_next && !next_same)
            HICOM_REQUIRE(hipStreamWaitEvent((hipStream_t)a.stream_next, (hipEvent_t)a.ev_done, 0) == hipSuccess, HICOM_ELAUNCH,
                          "compressor: stream wait");
    }
    if (merge_on_next)
        CHK(hicom_global_merge_fwd(F(w.part_m), F(w.part_l), F(w.part_acc), w.nparts, w.R, w.rows_pad, a.E, nullptr, 0, w.N,
                                   a.H, a.W, nullptr, 0, 0, 0, nullptr, ml_out, acc_out, solo ? 1 : 0, next_same ? sm : (hipStream_t)a.stream_next));
    return HICOM_OK;
}
