// Local compressor: windowed single-head cross-attention as ONE streaming pass.
//
// Replaces LocalCompressor.forward's divide_feature x3 + bmm + softmax + bmm + un-window
// (reference projector.py:544-558).  The window regroup is pure address arithmetic here
// (zero bytes moved); every key row and every value row is read from HBM exactly once.
//
// Mapping: one 256-thread workgroup (4 waves) per window.  A token row of D bf16 channels is
// spread over the 64 lanes of a wave as NV segments of 64 x 12 B (global_load_dwordx3, fully
// coalesced 768-B runs; the 3 tokens of a window row are contiguous in HBM).  Wave w scores
// tokens w, w+4, ... with a 64-lane shuffle reduction, the softmax over the <= few-hundred
// window scores is done in LDS, then the same wave->token assignment accumulates p * value in
// fp32 registers and the four partial rows are summed through LDS.
//
// Roofline: HBM-bound streaming (4*D flop per 4*D bytes); no MFMA on purpose -- one query row
// per window with private keys is a block-diagonal M=1 product (SURVEY.md §7 "Where MFMA applies").
#include "common.hpp"

namespace hicom {

struct __attribute__((packed, aligned(4))) Seg12 { uint32_t a, b, c; };

struct LocalParams {
    const void* key;
    const void* value;
    int key_f32, value_f32;      // stream dtype code: 0 = bf16 (raw tokens), 1 = fp32, 2 = fp16 (alpha-blended adaptor outputs)
    const void* query;
    int query_f32;
    long query_stride;
    hicom_axis at, ay, ax;
    float scale, bias;
    int l2norm;
    float* ctx;
    _Float16* ctx16;             // fp16 [Nw, D] (saturating) instead of / beside ctx: the operand of hicom_readout16_gemm_fwd, or NULL
};

template <int NV>
__device__ __forceinline__ void load_row(const uint16_t* row, int lane, float (&v)[NV][6]) {
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        const Seg12 g = *reinterpret_cast<const Seg12*>(row + 384 * s + 6 * lane);
        v[s][0] = bf16lo_to_f32(g.a); v[s][1] = bf16hi_to_f32(g.a);
        v[s][2] = bf16lo_to_f32(g.b); v[s][3] = bf16hi_to_f32(g.b);
        v[s][4] = bf16lo_to_f32(g.c); v[s][5] = bf16hi_to_f32(g.c);
    }
}

template <int NV>
__device__ __forceinline__ void load_row_f32(const float* row, int lane, float (&v)[NV][6]) {
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        const Seg12 a = *reinterpret_cast<const Seg12*>(row + 384 * s + 6 * lane);
        const Seg12 b = *reinterpret_cast<const Seg12*>(row + 384 * s + 6 * lane + 3);
        v[s][0] = __uint_as_float(a.a); v[s][1] = __uint_as_float(a.b); v[s][2] = __uint_as_float(a.c);
        v[s][3] = __uint_as_float(b.a); v[s][4] = __uint_as_float(b.b); v[s][5] = __uint_as_float(b.c);
    }
}

template <int NV>
__device__ __forceinline__ void load_row_f16(const _Float16* row, int lane, float (&v)[NV][6]) {
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        const Seg12 g = *reinterpret_cast<const Seg12*>(row + 384 * s + 6 * lane);
        const half2_t a = __builtin_bit_cast(half2_t, g.a), b = __builtin_bit_cast(half2_t, g.b), c = __builtin_bit_cast(half2_t, g.c);
        v[s][0] = (float)a[0]; v[s][1] = (float)a[1]; v[s][2] = (float)b[0]; v[s][3] = (float)b[1]; v[s][4] = (float)c[0]; v[s][5] = (float)c[1];
    }
}

template <int NV>
__device__ __forceinline__ void load_stream_row(const void* base, int dt, long token, int lane, float (&v)[NV][6]) {
    constexpr int D = NV * 384;
    if (dt == 1) load_row_f32<NV>(reinterpret_cast<const float*>(base) + token * D, lane, v);
    else if (dt == 2) load_row_f16<NV>(reinterpret_cast<const _Float16*>(base) + token * D, lane, v);
    else load_row<NV>(reinterpret_cast<const uint16_t*>(base) + token * D, lane, v);
}

template <int NV>
__global__ __launch_bounds__(256) void local_attn_kernel(LocalParams p) {
    constexpr int D = NV * 384;
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const int ks2 = p.ay.k * p.ax.k;
    const int WIN = p.at.k * ks2;
    float* sc = lsm;                              // [WIN] scores
    float* part = lsm + ((WIN + 3) & ~3);         // [4][D] partial contexts

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int win = blockIdx.x;
    const int w1 = win % p.ax.nwin;
    const int h1 = (win / p.ax.nwin) % p.ay.nwin;
    const int t1 = win / (p.ax.nwin * p.ay.nwin);
    const int t0 = axis_start(p.at, t1), y0 = axis_start(p.ay, h1), x0 = axis_start(p.ax, w1);
    const int H = p.ay.n, W = p.ax.n;

    // query fragment of this lane (fp32)
    float q[NV][6];
    if (p.query_f32) {
        const float* qp = reinterpret_cast<const float*>(p.query) + (long)win * p.query_stride;
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) q[s][j] = qp[384 * s + 6 * lane + j];
    } else {
        load_row<NV>(reinterpret_cast<const uint16_t*>(p.query) + (long)win * p.query_stride, lane, q);
    }
    if (p.l2norm & 2) {   // clip-scale variant: guide / ||guide||  (projector.py:529)
        float qq = 0.f;
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) qq += q[s][j] * q[s][j];
        const float inv = 1.0f / sqrtf(wave_sum_fast(qq));
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) q[s][j] *= inv;
    }

    auto token_of = [&](int i) -> long {
        const int t2 = i / ks2, r = i - t2 * ks2;
        const int h2 = r / p.ax.k, w2 = r - h2 * p.ax.k;
        return ((long)(t0 + t2) * H + (y0 + h2)) * W + (x0 + w2);
    };

    // ---- phase 1: scores ---------------------------------------------------------------
    for (int i0 = wave; i0 < WIN; i0 += 12) {     // up to 3 tokens in flight per wave
        float k[3][NV][6];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) load_stream_row<NV>(p.key, p.key_f32, token_of(i), lane, k[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                float dot = 0.f, kk = 0.f;
#pragma unroll
                for (int s = 0; s < NV; ++s)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        dot = fmaf(q[s][j], k[u][s][j], dot);
                        kk = fmaf(k[u][s][j], k[u][s][j], kk);
                    }
                dot = wave_sum_fast(dot);
                if (p.l2norm & 1) dot /= sqrtf(wave_sum_fast(kk));   // frames_embed / ||.||  (:528)
                if (lane == 0) sc[i] = dot * p.scale + p.bias;
            }
        }
    }
    __syncthreads();

    // ---- softmax statistics (every wave redundantly; WIN is tiny) -------------------------
    float mx = -3.0e38f;
    for (int i = lane; i < WIN; i += 64) mx = fmaxf(mx, sc[i]);
    mx = wave_max_fast(mx);
    float sum = 0.f;
    for (int i = lane; i < WIN; i += 64) sum += expf(sc[i] - mx);
    const float inv_sum = 1.0f / wave_sum_fast(sum);

    // ---- phase 2: context = sum_i p_i * value_i ---------------------------------------------
    float acc[NV][6];
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[s][j] = 0.f;
    for (int i0 = wave; i0 < WIN; i0 += 12) {
        float v[3][NV][6];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) load_stream_row<NV>(p.value, p.value_f32, token_of(i), lane, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                const float pi = expf(sc[i] - mx) * inv_sum;
#pragma unroll
                for (int s = 0; s < NV; ++s)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[s][j] = fmaf(pi, v[u][s][j], acc[s][j]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) part[wave * D + 384 * s + 6 * lane + j] = acc[s][j];
    __syncthreads();
    for (int c = tid; c < D; c += 256) {
        const float v = (part[c] + part[D + c]) + (part[2 * D + c] + part[3 * D + c]);
        if (p.ctx) p.ctx[(long)win * D + c] = v;
        if (p.ctx16) p.ctx16[(long)win * D + c] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
    }
}

// ---- windowed attention with the adaptor blend fused into the row loads ("fused into the K/V tile load") -------
// adapt_k / adapt_v (reference projector.py:533-534):  key_n = (1 - a_k) x_n + a_k (LN(y_n) gamma_k + beta_k)  with y = k_proj(x) from the
// two dense GEMMs, likewise for the values.  The blended streams never exist: a wave that holds a token's x row and y row computes
//   logit_n = (1 - a) q.x_n + a (rstd_n ((q gamma).y_n - mu_n sum(q gamma)) + q.beta)                      (four wave reductions)
//   ctx     = (1 - a) sum_n p_n v_n + a (gamma (sum_n p_n rstd_n y_n - sum_n p_n rstd_n mu_n) + beta)       (two accumulators)
// with mu_n, rstd_n the LayerNorm statistics of y_n over D (fp32, from the row in registers).  Saves, per adapted stream, the
// LayerNorm-blend pass (107 MB fp16 + 107 MB bf16 read, 107 MB written) and costs one more 107-MB row read here.
struct LocalAdaptParams {
    const uint16_t* kx;        // key source tokens bf16 [N, D]
    const _Float16* ky;        // k_proj(kx) fp16 [N, D] or NULL (no key adaptor: key = kx)
    const uint16_t* kgamma; const uint16_t* kbeta; const void* kalpha;
    const uint16_t* vx;        // value source tokens bf16 [N, D]
    const _Float16* vy;        // v_proj(vx) fp16 [N, D] or NULL
    const uint16_t* vgamma; const uint16_t* vbeta; const void* valpha;
    int alpha_f32;
    float eps;
    const void* query;
    int query_f32;
    long query_stride;
    hicom_axis at, ay, ax;
    float scale, bias;
    float* ctx;
};

template <int NV>
__global__ __launch_bounds__(256) void local_attn_adapt_kernel(LocalAdaptParams p) {
    constexpr int D = NV * 384;
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const int ks2 = p.ay.k * p.ax.k;
    const int WIN = p.at.k * ks2;
    float* sc = lsm;                              // [WIN] scores
    float* part = lsm + ((WIN + 3) & ~3);         // [4][D] partial contexts (x part, then y part)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int win = blockIdx.x;
    const int w1 = win % p.ax.nwin;
    const int h1 = (win / p.ax.nwin) % p.ay.nwin;
    const int t1 = win / (p.ax.nwin * p.ay.nwin);
    const int t0 = axis_start(p.at, t1), y0 = axis_start(p.ay, h1), x0 = axis_start(p.ax, w1);
    const int H = p.ay.n, W = p.ax.n;
    auto scalar = [&](const void* a) { return p.alpha_f32 ? *reinterpret_cast<const float*>(a) : bf16_to_f32(*reinterpret_cast<const uint16_t*>(a)); };
    const float ak = p.ky ? scalar(p.kalpha) : 0.f, av = p.vy ? scalar(p.valpha) : 0.f;

    float q[NV][6];
    if (p.query_f32) {
        const float* qp = reinterpret_cast<const float*>(p.query) + (long)win * p.query_stride;
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) q[s][j] = qp[384 * s + 6 * lane + j];
    } else {
        load_row<NV>(reinterpret_cast<const uint16_t*>(p.query) + (long)win * p.query_stride, lane, q);
    }
    // q gamma_k (the query seen by the normalised y row), sum(q gamma_k), q . beta_k
    float qg[NV][6];
    float cg = 0.f, cb = 0.f;
    if (p.ky) {
        float g[NV][6], b[NV][6];
        load_row<NV>(p.kgamma, lane, g);
        load_row<NV>(p.kbeta, lane, b);
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                qg[s][j] = q[s][j] * g[s][j];
                cg += qg[s][j];
                cb = fmaf(q[s][j], b[s][j], cb);
            }
        cg = wave_sum_fast(cg);
        cb = wave_sum_fast(cb);
    }
    auto token_of = [&](int i) -> long {
        const int t2 = i / ks2, r = i - t2 * ks2;
        const int h2 = r / p.ax.k, w2 = r - h2 * p.ax.k;
        return ((long)(t0 + t2) * H + (y0 + h2)) * W + (x0 + w2);
    };

    // ---- phase 1: scores ---------------------------------------------------------------
    for (int i0 = wave; i0 < WIN; i0 += 8) {      // 2 tokens (x row + y row each) in flight per wave
        float kx[2][NV][6], ky[2][NV][6];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                const long tok = token_of(i);
                load_row<NV>(p.kx + tok * D, lane, kx[u]);
                if (p.ky) load_row_f16<NV>(p.ky + tok * D, lane, ky[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                float dx = 0.f, dy = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int s = 0; s < NV; ++s)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        dx = fmaf(q[s][j], kx[u][s][j], dx);
                        if (p.ky) {
                            dy = fmaf(qg[s][j], ky[u][s][j], dy);
                            s1 += ky[u][s][j];
                        }
                    }
                dx = wave_sum_fast(dx);
                float logit = dx;
                if (p.ky) {
                    dy = wave_sum_fast(dy);
                    const float mu = wave_sum_fast(s1) * (1.0f / D);
#pragma unroll
                    for (int s = 0; s < NV; ++s)
#pragma unroll
                        for (int j = 0; j < 6; ++j) { const float d = ky[u][s][j] - mu; s2 = fmaf(d, d, s2); }
                    const float rstd = 1.0f / sqrtf(wave_sum_fast(s2) * (1.0f / D) + p.eps);
                    logit = (1.0f - ak) * dx + ak * (rstd * (dy - mu * cg) + cb);
                }
                if (lane == 0) sc[i] = logit * p.scale + p.bias;
            }
        }
    }
    __syncthreads();
    float mx = -3.0e38f;
    for (int i = lane; i < WIN; i += 64) mx = fmaxf(mx, sc[i]);
    mx = wave_max_fast(mx);
    float sum = 0.f;
    for (int i = lane; i < WIN; i += 64) sum += expf(sc[i] - mx);
    const float inv_sum = 1.0f / wave_sum_fast(sum);

    // ---- phase 2: context ----------------------------------------------------------------
    float accx[NV][6], accy[NV][6];
    float smu = 0.f;                                // sum_n p_n rstd_n mu_n (identical in every lane)
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) { accx[s][j] = 0.f; accy[s][j] = 0.f; }
    for (int i0 = wave; i0 < WIN; i0 += 8) {
        float vx[2][NV][6], vy[2][NV][6];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                const long tok = token_of(i);
                load_row<NV>(p.vx + tok * D, lane, vx[u]);
                if (p.vy) load_row_f16<NV>(p.vy + tok * D, lane, vy[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                const float pi = expf(sc[i] - mx) * inv_sum;
                float w = 0.f;
                if (p.vy) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int s = 0; s < NV; ++s)
#pragma unroll
                        for (int j = 0; j < 6; ++j) s1 += vy[u][s][j];
                    const float mu = wave_sum_fast(s1) * (1.0f / D);
#pragma unroll
                    for (int s = 0; s < NV; ++s)
#pragma unroll
                        for (int j = 0; j < 6; ++j) { const float d = vy[u][s][j] - mu; s2 = fmaf(d, d, s2); }
                    const float rstd = 1.0f / sqrtf(wave_sum_fast(s2) * (1.0f / D) + p.eps);
                    w = pi * rstd;
                    smu = fmaf(w, mu, smu);
                }
#pragma unroll
                for (int s = 0; s < NV; ++s)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        accx[s][j] = fmaf(pi, vx[u][s][j], accx[s][j]);
                        if (p.vy) accy[s][j] = fmaf(w, vy[u][s][j], accy[s][j]);
                    }
            }
        }
    }
    // per-wave blend (linear in the accumulators, so the four waves' results simply add: the beta term once, by wave 0)
    float out_[NV][6];
    if (p.vy) {
        float g[NV][6], b[NV][6];
        load_row<NV>(p.vgamma, lane, g);
        load_row<NV>(p.vbeta, lane, b);
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                out_[s][j] = (1.0f - av) * accx[s][j] + av * (g[s][j] * (accy[s][j] - smu) + (wave == 0 ? b[s][j] : 0.f));
    } else {
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) out_[s][j] = accx[s][j];
    }
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) part[wave * D + 384 * s + 6 * lane + j] = out_[s][j];
    __syncthreads();
    float* out = p.ctx + (long)win * D;
    for (int c = tid; c < D; c += 256) out[c] = (part[c] + part[D + c]) + (part[2 * D + c] + part[3 * D + c]);
}

// ---- backward of the windowed attention (training path; stage 3 of the reference's script trains the
// SigLIP head and the guide encoder too: train.py:717-726, so the gradients w.r.t. the KEY stream frames_embed and the query
// are needed; frames_feature comes from the frozen tower body).  Per window, autograd through projector.py:550-553:
//     s_i = scale q.k_i + bias,  p = softmax(s),  ctx = sum_i p_i v_i
//     dP_i = dctx . v_i,  dS_i = p_i (dP_i - sum_j p_j dP_j),  dq = scale sum_i dS_i k_i,  dk_i = scale dS_i q
// Same mapping as the forward kernel: one 4-wave workgroup per window, a token row over the 64 lanes; the key rows are read
// twice (scores, then dq), the value rows once.  dkey is written by plain stores: the caller guarantees an exact partition
// (every token in exactly one window).
struct LocalBwdParams {
    const uint16_t* key;
    const uint16_t* value;
    const void* query;
    int query_f32;
    long query_stride;
    hicom_axis at, ay, ax;
    float scale, bias;
    const float* dctx;     // [Nw, D]
    float* dq;             // [Nw, D]
    uint16_t* dkey;        // bf16 [N, D] or NULL
    // clip-scale (reference projector.py:527-529, :549; round 6): the key rows are L2-normalised, khat_i = k_i / ||k_i||, and
    // s_i = e^ls (q . khat_i) + lb  (scale = e^ls, bias = lb).  Then  dq = scale sum_i dS_i khat_i,
    // dk_i = scale dS_i (q - khat_i (q . khat_i)) / ||k_i||  (through the normalisation), and -- since ds_i / d ls = s_i - lb --
    // dls[w] = sum_i dS_i (s_i - lb) per window (d lb = sum_i dS_i = 0 exactly: the softmax cancels a shift).
    int l2norm_key;
    float* dls;            // [Nw] or NULL
    // d VALUE stream (round 6: d frames_feature, reference train.py:712-715 `pure_vision_model` trains the tower body): dv_i = p_i dctx_w,
    // bf16 [N, D], plain stores (exact partition) -- or NULL.  value_is_key (frames_embed is None: the keys ARE the value rows,
    // projector.py:532): the key-side gradient of the same row is added in, d x_i = p_i dctx_w + dk_i.
    uint16_t* dvalue;
    int value_is_key;
    // Overlapping windows (an axis the kernel does not divide: reference projector.py:501-522 shifts the trailing windows by k - 1, so a window
    // shares ONE plane of tokens with its predecessor): the per-token outputs are then ACCUMULATED (read-modify-write of the bf16 rows, cleared by
    // the host), one launch per parity class of the window index along every such axis -- windows of one class never share a token, so the
    // sums are ordered by the launches: deterministic, no atomics.  par_mask: bit a set = axis a (t, y, x) is filtered; par_val: the class.
    int accumulate, par_mask, par_val;
};

__device__ __forceinline__ void seg12_add(float (&e)[6], const uint16_t* o) {
    const Seg12 w = *reinterpret_cast<const Seg12*>(o);
    e[0] += bf16lo_to_f32(w.a); e[1] += bf16hi_to_f32(w.a); e[2] += bf16lo_to_f32(w.b); e[3] += bf16hi_to_f32(w.b);
    e[4] += bf16lo_to_f32(w.c); e[5] += bf16hi_to_f32(w.c);
}

template <int NV>
__global__ __launch_bounds__(256) void local_attn_bwd_kernel(LocalBwdParams p) {
    constexpr int D = NV * 384;
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const int ks2 = p.ay.k * p.ax.k;
    const int WIN = p.at.k * ks2;
    const int WP = (WIN + 3) & ~3;
    float* sc = lsm;                              // [WIN] scores, then dS
    float* dp = lsm + WP;                         // [WIN] dP
    float* rin = lsm + 2 * WP;                    // [WIN] 1 / ||k_i|| (1 without l2norm_key)
    float* sraw = lsm + 3 * WP;                   // [WIN] q . khat_i
    float* pwt = lsm + 4 * WP;                    // [WIN] softmax weights p_i
    float* part = lsm + 5 * WP;                   // [4][D] partial dq

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int win = blockIdx.x;
    const int w1 = win % p.ax.nwin;
    const int h1 = (win / p.ax.nwin) % p.ay.nwin;
    const int t1 = win / (p.ax.nwin * p.ay.nwin);
    const int t0 = axis_start(p.at, t1), y0 = axis_start(p.ay, h1), x0 = axis_start(p.ax, w1);
    const int H = p.ay.n, W = p.ax.n;
    if ((((t1 & 1) | ((h1 & 1) << 1) | ((w1 & 1) << 2)) & p.par_mask) != p.par_val) return;      // (another launch's parity class)

    float q[NV][6], g[NV][6];
    if (p.query_f32) {
        const float* qp = reinterpret_cast<const float*>(p.query) + (long)win * p.query_stride;
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) q[s][j] = qp[384 * s + 6 * lane + j];
    } else {
        load_row<NV>(reinterpret_cast<const uint16_t*>(p.query) + (long)win * p.query_stride, lane, q);
    }
    load_row_f32<NV>(p.dctx + (long)win * D, lane, g);

    auto token_of = [&](int i) -> long {
        const int t2 = i / ks2, r = i - t2 * ks2;
        const int h2 = r / p.ax.k, w2 = r - h2 * p.ax.k;
        return ((long)(t0 + t2) * H + (y0 + h2)) * W + (x0 + w2);
    };

    // ---- phase 1: scores and dP ------------------------------------------------------------------
    for (int i0 = wave; i0 < WIN; i0 += 8) {      // 2 tokens (key + value rows) in flight per wave
        float k[2][NV][6], v[2][NV][6];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                const long tok = token_of(i);
                load_row<NV>(p.key + tok * D, lane, k[u]);
                load_row<NV>(p.value + tok * D, lane, v[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                float dot = 0.f, dd = 0.f, kk = 0.f;
#pragma unroll
                for (int s = 0; s < NV; ++s)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        dot = fmaf(q[s][j], k[u][s][j], dot);
                        dd = fmaf(g[s][j], v[u][s][j], dd);
                        kk = fmaf(k[u][s][j], k[u][s][j], kk);
                    }
                dot = wave_sum_fast(dot);
                dd = wave_sum_fast(dd);
                float ri = 1.0f;
                if (p.l2norm_key) {
                    ri = 1.0f / sqrtf(wave_sum_fast(kk));
                    dot *= ri;
                }
                if (lane == 0) { sc[i] = dot * p.scale + p.bias; dp[i] = dd; rin[i] = ri; sraw[i] = dot; }
            }
        }
    }
    __syncthreads();

    // ---- softmax statistics and delta (every wave redundantly; WIN is tiny) ----------------------------
    float mx = -3.0e38f;
    for (int i = lane; i < WIN; i += 64) mx = fmaxf(mx, sc[i]);
    mx = wave_max_fast(mx);
    float sum = 0.f, pd = 0.f;
    for (int i = lane; i < WIN; i += 64) {
        const float e = expf(sc[i] - mx);
        sum += e;
        pd = fmaf(e, dp[i], pd);
    }
    const float inv_sum = 1.0f / wave_sum_fast(sum);
    const float delta = wave_sum_fast(pd) * inv_sum;
    __syncthreads();                               // every wave has read sc / dp as scores
    if (wave == 0) {
        float dl = 0.f;
        for (int i = lane; i < WIN; i += 64) {
            const float s_i = sc[i], pi_ = expf(s_i - mx) * inv_sum, dS = pi_ * (dp[i] - delta);
            dl = fmaf(dS, s_i - p.bias, dl);
            sc[i] = dS;
            pwt[i] = pi_;
        }
        if (p.dls) {
            dl = wave_sum_fast(dl);
            if (lane == 0) p.dls[win] = dl;
        }
    }
    __syncthreads();

    // ---- phase 2: dq = scale sum_i dS_i k_i ;  dkey_i = scale dS_i q -----------------------------------------
    float acc[NV][6];
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[s][j] = 0.f;
    for (int i0 = wave; i0 < WIN; i0 += 12) {
        float k[3][NV][6];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) load_row<NV>(p.key + token_of(i) * D, lane, k[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = i0 + 4 * u;
            if (i < WIN) {
                // (plain: dsk = ds, kc = 0.  clip-scale: d k_i = ds / ||k|| (q - khat (q . khat)) = dsk q - kc k with dsk = ds / ||k||,
                // kc = dsk (q . khat) / ||k||; the dq sum runs over khat = k / ||k||)
                const float ds = sc[i] * p.scale, dsk = ds * rin[i], kc = p.l2norm_key ? dsk * sraw[i] * rin[i] : 0.f;
#pragma unroll
                for (int s = 0; s < NV; ++s)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[s][j] = fmaf(dsk, k[u][s][j], acc[s][j]);
                if (p.dkey) {
                    uint16_t* o = p.dkey + token_of(i) * D;
#pragma unroll
                    for (int s = 0; s < NV; ++s) {
                        float e[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) e[j] = fmaf(dsk, q[s][j], -kc * k[u][s][j]);
                        if (p.accumulate) seg12_add(e, o + 384 * s + 6 * lane);
                        Seg12 w;
                        w.a = f32_to_bf16(e[0]) | ((uint32_t)f32_to_bf16(e[1]) << 16);
                        w.b = f32_to_bf16(e[2]) | ((uint32_t)f32_to_bf16(e[3]) << 16);
                        w.c = f32_to_bf16(e[4]) | ((uint32_t)f32_to_bf16(e[5]) << 16);
                        *reinterpret_cast<Seg12*>(o + 384 * s + 6 * lane) = w;
                    }
                }
                if (p.dvalue) {
                    uint16_t* o = p.dvalue + token_of(i) * D;
                    const float pi_ = pwt[i];
#pragma unroll
                    for (int s = 0; s < NV; ++s) {
                        float e[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            e[j] = pi_ * g[s][j];
                            if (p.value_is_key) e[j] += fmaf(dsk, q[s][j], -kc * k[u][s][j]);
                        }
                        if (p.accumulate) seg12_add(e, o + 384 * s + 6 * lane);
                        Seg12 w;
                        w.a = f32_to_bf16(e[0]) | ((uint32_t)f32_to_bf16(e[1]) << 16);
                        w.b = f32_to_bf16(e[2]) | ((uint32_t)f32_to_bf16(e[3]) << 16);
                        w.c = f32_to_bf16(e[4]) | ((uint32_t)f32_to_bf16(e[5]) << 16);
                        *reinterpret_cast<Seg12*>(o + 384 * s + 6 * lane) = w;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) part[wave * D + 384 * s + 6 * lane + j] = acc[s][j];
    __syncthreads();
    float* out = p.dq + (long)win * D;
    for (int c = tid; c < D; c += 256) out[c] = (part[c] + part[D + c]) + (part[2 * D + c] + part[3 * D + c]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the windowed attention WITH the k / v adaptor blends (training path of the second released recipe
// `local43_adaptkv_global32`; autograd through reference projector.py:533-534 + :550-553).  Round 3 ran this half as fp32 torch
// algebra over window-regrouped [1296, 36, 1152] tensors; here it is two streaming passes over the four token streams
// (x_k, y_k = k_proj(x_k), x_v, y_v), same mapping as the forward kernel (one 4-wave workgroup per window, a token row over a wave):
//     K_n = (1 - a_k) x_n + a_k (gamma_k yhat_n + beta_k),   yhat_n = (y_n - mu_n) rstd_n          (likewise V_n)
//     s_n = scale q.K_n + bias,  p = softmax(s),  dP_n = dctx . V_n,  dS_n = p_n (dP_n - sum_m p_m dP_m)
// Outputs -- everything the rest of the backward needs, none of it of token-stream size except two scalars per token:
//     ds[tok] = scale dS_n,  pw[tok] = p_n                     (d K_n = ds q,  d V_n = pw dctx: rank-1 per token, never materialised)
//     sxk[w] = sum_n ds_n x_k,n    syk[w] = sum_n ds_n yhat_k,n    sxv[w] = sum_n p_n x_v,n    syv[w] = sum_n p_n yhat_v,n      ([Nw, D] f32)
// from which dq = (1 - a_k) sxk + a_k gamma_k syk, d alpha, d gamma, d beta of both adaptors follow as reductions over the windows
// (tiny), and the adaptor-MLP backward starts from (ds, q) / (pw, dctx) through hicom_adapt_dy_fwd.
struct LocalAdaptBwdParams {
    const uint16_t* kx; const _Float16* ky; const uint16_t* kgamma; const uint16_t* kbeta; const void* kalpha;
    const uint16_t* vx; const _Float16* vy; const uint16_t* vgamma; const uint16_t* vbeta; const void* valpha;
    int alpha_f32;
    float eps;
    const void* query;
    int query_f32;
    long query_stride;
    hicom_axis at, ay, ax;
    float scale, bias;
    const float* dctx;      // [Nw, D]
    float* ds;              // [N] token-indexed
    float* pw;              // [N]
    float* sxk; float* syk; float* sxv; float* syv;     // [Nw, D] (syk / syv NULL when that stream has no adaptor)
};

template <int NV>
__global__ __launch_bounds__(256) void local_attn_adapt_bwd_kernel(LocalAdaptBwdParams p) {
    constexpr int D = NV * 384;
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const int ks2 = p.ay.k * p.ax.k;
    const int WIN = p.at.k * ks2;
    const int WP = (WIN + 3) & ~3;
    float* sc = lsm;                               // [WIN] scores, then ds
    float* dp = lsm + WP;                          // [WIN] dP, then p
    float* stk = lsm + 2 * WP;                     // [2][WIN] mu, rstd of y_k
    float* stv = lsm + 4 * WP;                     // [2][WIN] mu, rstd of y_v
    float* part = lsm + 6 * WP;                    // [4][D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int win = blockIdx.x;
    const int w1 = win % p.ax.nwin;
    const int h1 = (win / p.ax.nwin) % p.ay.nwin;
    const int t1 = win / (p.ax.nwin * p.ay.nwin);
    const int t0 = axis_start(p.at, t1), y0 = axis_start(p.ay, h1), x0 = axis_start(p.ax, w1);
    const int H = p.ay.n, W = p.ax.n;
    auto scalar = [&](const void* a) { return p.alpha_f32 ? *reinterpret_cast<const float*>(a) : bf16_to_f32(*reinterpret_cast<const uint16_t*>(a)); };
    const float ak = p.ky ? scalar(p.kalpha) : 0.f, av = p.vy ? scalar(p.valpha) : 0.f;

    float q[NV][6], g[NV][6];
    if (p.query_f32) {
        const float* qp = reinterpret_cast<const float*>(p.query) + (long)win * p.query_stride;
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) q[s][j] = qp[384 * s + 6 * lane + j];
    } else {
        load_row<NV>(reinterpret_cast<const uint16_t*>(p.query) + (long)win * p.query_stride, lane, q);
    }
    load_row_f32<NV>(p.dctx + (long)win * D, lane, g);
    // q gamma_k, sum(q gamma_k), q . beta_k  and  g gamma_v, sum(g gamma_v), g . beta_v
    float qg[NV][6], gg[NV][6];
    float cgk = 0.f, cbk = 0.f, cgv = 0.f, cbv = 0.f;
    if (p.ky) {
        float ga[NV][6], be[NV][6];
        load_row<NV>(p.kgamma, lane, ga);
        load_row<NV>(p.kbeta, lane, be);
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) { qg[s][j] = q[s][j] * ga[s][j]; cgk += qg[s][j]; cbk = fmaf(q[s][j], be[s][j], cbk); }
        cgk = wave_sum_fast(cgk);
        cbk = wave_sum_fast(cbk);
    }
    if (p.vy) {
        float ga[NV][6], be[NV][6];
        load_row<NV>(p.vgamma, lane, ga);
        load_row<NV>(p.vbeta, lane, be);
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) { gg[s][j] = g[s][j] * ga[s][j]; cgv += gg[s][j]; cbv = fmaf(g[s][j], be[s][j], cbv); }
        cgv = wave_sum_fast(cgv);
        cbv = wave_sum_fast(cbv);
    }
    auto token_of = [&](int i) -> long {
        const int t2 = i / ks2, r = i - t2 * ks2;
        const int h2 = r / p.ax.k, w2 = r - h2 * p.ax.k;
        return ((long)(t0 + t2) * H + (y0 + h2)) * W + (x0 + w2);
    };
    // blended dot product of one stream's token with a (vector, vector * gamma) pair; also leaves the LayerNorm statistics
    auto blend_dot = [&](const float (&x)[NV][6], const float (&y)[NV][6], bool has_y, const float (&v)[NV][6], const float (&vgm)[NV][6], float cg,
                         float cb, float a, float& mu_o, float& rstd_o) -> float {
        float dx = 0.f, dy = 0.f, s1 = 0.f;
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                dx = fmaf(v[s][j], x[s][j], dx);
                if (has_y) { dy = fmaf(vgm[s][j], y[s][j], dy); s1 += y[s][j]; }
            }
        dx = wave_sum_fast(dx);
        if (!has_y) return dx;
        dy = wave_sum_fast(dy);
        const float mu = wave_sum_fast(s1) * (1.0f / D);
        float s2 = 0.f;
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) { const float d = y[s][j] - mu; s2 = fmaf(d, d, s2); }
        const float rstd = 1.0f / sqrtf(wave_sum_fast(s2) * (1.0f / D) + p.eps);
        mu_o = mu; rstd_o = rstd;
        return (1.0f - a) * dx + a * (rstd * (dy - mu * cg) + cb);
    };

    // ---- phase 1: scores, dP, LayerNorm statistics (one token per wave in flight: four rows) -----------------------------------
    for (int i = wave; i < WIN; i += 4) {
        const long tok = token_of(i);
        float kx[NV][6], ky[NV][6], vx[NV][6], vy[NV][6];
        load_row<NV>(p.kx + tok * D, lane, kx);
        if (p.ky) load_row_f16<NV>(p.ky + tok * D, lane, ky);
        load_row<NV>(p.vx + tok * D, lane, vx);
        if (p.vy) load_row_f16<NV>(p.vy + tok * D, lane, vy);
        float muk = 0.f, rsk = 0.f, muv = 0.f, rsv = 0.f;
        const float logit = blend_dot(kx, ky, p.ky != nullptr, q, qg, cgk, cbk, ak, muk, rsk);
        const float dpi = blend_dot(vx, vy, p.vy != nullptr, g, gg, cgv, cbv, av, muv, rsv);
        if (lane == 0) {
            sc[i] = logit * p.scale + p.bias;
            dp[i] = dpi;
            stk[i] = muk; stk[WP + i] = rsk;
            stv[i] = muv; stv[WP + i] = rsv;
        }
    }
    __syncthreads();
    // ---- softmax statistics and delta (every wave redundantly; WIN is tiny) ----------------------------------------------------
    float mx = -3.0e38f;
    for (int i = lane; i < WIN; i += 64) mx = fmaxf(mx, sc[i]);
    mx = wave_max_fast(mx);
    float sum = 0.f, pd = 0.f;
    for (int i = lane; i < WIN; i += 64) {
        const float e = expf(sc[i] - mx);
        sum += e;
        pd = fmaf(e, dp[i], pd);
    }
    const float inv_sum = 1.0f / wave_sum_fast(sum);
    const float delta = wave_sum_fast(pd) * inv_sum;
    __syncthreads();                               // every wave has read sc / dp as scores / dP
    if (wave == 0)
        for (int i = lane; i < WIN; i += 64) {
            const float pi = expf(sc[i] - mx) * inv_sum;
            const float dsi = pi * (dp[i] - delta) * p.scale;
            sc[i] = dsi;
            dp[i] = pi;
            const long tok = token_of(i);
            p.ds[tok] = dsi;
            p.pw[tok] = pi;
        }
    __syncthreads();
    // ---- phase 2: the four weighted sums over the window ------------------------------------------------------------------------
    float axk[NV][6], ayk[NV][6], axv[NV][6], ayv[NV][6];
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) { axk[s][j] = 0.f; ayk[s][j] = 0.f; axv[s][j] = 0.f; ayv[s][j] = 0.f; }
    for (int i = wave; i < WIN; i += 4) {
        const long tok = token_of(i);
        float kx[NV][6], ky[NV][6], vx[NV][6], vy[NV][6];
        load_row<NV>(p.kx + tok * D, lane, kx);
        if (p.ky) load_row_f16<NV>(p.ky + tok * D, lane, ky);
        load_row<NV>(p.vx + tok * D, lane, vx);
        if (p.vy) load_row_f16<NV>(p.vy + tok * D, lane, vy);
        const float dsi = sc[i], pi = dp[i];
        const float muk = stk[i], wk = dsi * stk[WP + i], muv = stv[i], wv = pi * stv[WP + i];
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                axk[s][j] = fmaf(dsi, kx[s][j], axk[s][j]);
                axv[s][j] = fmaf(pi, vx[s][j], axv[s][j]);
                if (p.ky) ayk[s][j] = fmaf(wk, ky[s][j] - muk, ayk[s][j]);
                if (p.vy) ayv[s][j] = fmaf(wv, vy[s][j] - muv, ayv[s][j]);
            }
    }
    auto reduce_out = [&](const float (&a)[NV][6], float* dst) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) part[wave * D + 384 * s + 6 * lane + j] = a[s][j];
        __syncthreads();
        float* out = dst + (long)win * D;
        for (int c = tid; c < D; c += 256) out[c] = (part[c] + part[D + c]) + (part[2 * D + c] + part[3 * D + c]);
    };
    reduce_out(axk, p.sxk);
    reduce_out(axv, p.sxv);
    if (p.ky) reduce_out(ayk, p.syk);
    if (p.vy) reduce_out(ayv, p.syv);
}

// ---- d y of one adaptor stream: backward of the LayerNorm blend for a rank-1 upstream gradient ---------------------------------------
// The gradient w.r.t. the blended row is coef[tok] * vec[w(tok)] (K stream: ds * q, V stream: pw * dctx), so with n = gamma yhat + beta,
// g = d yhat = alpha coef (vec gamma):  dy = rstd (g - mean(g) - yhat mean(g yhat)).  One wave per token: reads the fp16 y row, writes dy as
// bf16 (the operand dtype of the GEMMs behind it); optionally also r1[tok] = coef2 coef[tok] vec[w] (the (1 - alpha) x-branch of d
// frames_embed, stage 3).  Exact window partition (every token in one window).
struct AdaptDyParams {
    const _Float16* y;      // [N, D]
    const uint16_t* gamma;
    const void* vec;        // f32 | bf16 [Nw | 1, D]
    int vec_f32;
    long vec_stride;
    const float* coef;      // [N]
    const void* alpha; int alpha_f32;
    float eps;
    hicom_axis at, ay, ax;
    uint16_t* dy;           // bf16 [N, D]
    uint16_t* r1;           // bf16 [N, D] or NULL
    long N;
    float* col_parts;       // f32 [gridDim.x][D] or NULL: this workgroup's column sums of dy (as stored: bf16-rounded) -- the bias gradient
};

template <int NV>
__global__ __launch_bounds__(256) void adapt_dy_kernel(AdaptDyParams p) {
    constexpr int D = NV * 384;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float csum[NV][6];
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) csum[s][j] = 0.f;
    for (long tok = (long)blockIdx.x * 4 + wave; tok < p.N; tok += (long)gridDim.x * 4) {
    const int H = p.ay.n, W = p.ax.n;
    const int t = (int)(tok / ((long)H * W)), rem = (int)(tok - (long)t * H * W), yy = rem / W, xx = rem - yy * W;
    const long win = ((long)(t / p.at.k) * p.ay.nwin + yy / p.ay.k) * p.ax.nwin + xx / p.ax.k;
    const float alpha = p.alpha_f32 ? *reinterpret_cast<const float*>(p.alpha) : bf16_to_f32(*reinterpret_cast<const uint16_t*>(p.alpha));
    const float c = p.coef[tok];
    float y[NV][6], v[NV][6], ga[NV][6];
    load_row_f16<NV>(p.y + tok * D, lane, y);
    if (p.vec_f32) load_row_f32<NV>(reinterpret_cast<const float*>(p.vec) + win * p.vec_stride, lane, v);
    else load_row<NV>(reinterpret_cast<const uint16_t*>(p.vec) + win * p.vec_stride, lane, v);
    load_row<NV>(p.gamma, lane, ga);
    float s1 = 0.f;
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) s1 += y[s][j];
    const float mu = wave_sum_fast(s1) * (1.0f / D);
    float s2 = 0.f;
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) { y[s][j] -= mu; s2 = fmaf(y[s][j], y[s][j], s2); }
    const float rstd = 1.0f / sqrtf(wave_sum_fast(s2) * (1.0f / D) + p.eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            y[s][j] *= rstd;                         // yhat
            const float gv = v[s][j] * ga[s][j];
            m1 += gv;
            m2 = fmaf(gv, y[s][j], m2);
            ga[s][j] = gv;                           // (vec gamma)
        }
    m1 = wave_sum_fast(m1) * (1.0f / D);
    m2 = wave_sum_fast(m2) * (1.0f / D);
    const float k = alpha * c * rstd;
    uint16_t* o = p.dy + tok * D;
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        float r[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) r[j] = k * (ga[s][j] - m1 - y[s][j] * m2);
        Seg12 w;
        w.a = f32_to_bf16(r[0]) | ((uint32_t)f32_to_bf16(r[1]) << 16);
        w.b = f32_to_bf16(r[2]) | ((uint32_t)f32_to_bf16(r[3]) << 16);
        w.c = f32_to_bf16(r[4]) | ((uint32_t)f32_to_bf16(r[5]) << 16);
        *reinterpret_cast<Seg12*>(o + 384 * s + 6 * lane) = w;
        csum[s][0] += bf16lo_to_f32(w.a); csum[s][1] += bf16hi_to_f32(w.a); csum[s][2] += bf16lo_to_f32(w.b);
        csum[s][3] += bf16hi_to_f32(w.b); csum[s][4] += bf16lo_to_f32(w.c); csum[s][5] += bf16hi_to_f32(w.c);
    }
    if (p.r1) {
        const float k1 = (1.0f - alpha) * c;
        uint16_t* o1 = p.r1 + tok * D;
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            Seg12 w;
            w.a = f32_to_bf16(k1 * v[s][0]) | ((uint32_t)f32_to_bf16(k1 * v[s][1]) << 16);
            w.b = f32_to_bf16(k1 * v[s][2]) | ((uint32_t)f32_to_bf16(k1 * v[s][3]) << 16);
            w.c = f32_to_bf16(k1 * v[s][4]) | ((uint32_t)f32_to_bf16(k1 * v[s][5]) << 16);
            *reinterpret_cast<Seg12*>(o1 + 384 * s + 6 * lane) = w;
        }
    }
    }   // token loop
    if (p.col_parts) {
        __shared__ float red[4][D];
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) red[wave][384 * s + 6 * lane + j] = csum[s][j];
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) p.col_parts[(long)blockIdx.x * D + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// ---- elementwise helpers of the adaptor-MLP backward (token-stream sized, 16-byte accesses) ------------------------------------------------
// erf-GELU derivative Phi(x) + x phi(x), same erf approximation as gelu_erf (A&S 7.1.26)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);           // exp(-x^2 / 2)
    const float tail = poly * t * e;                                                    // 1 - erf(z)
    const float cdf = x >= 0.f ? 1.0f - 0.5f * tail : 0.5f * tail;
    return fmaf(x * 0.3989422804014327f, e, cdf);
}

// tanh form (HF gelu_pytorch_tanh, the SigLIP head's activation) and its derivative:  g(x) = x s(2u),  u = c0 (x + c1 x^3),
// s = logistic;  g'(x) = s + x s (1 - s) 2 c0 (1 + 3 c1 x^2)
__device__ __forceinline__ float gelu_tanh_val(float x) {
    const float t = x * x;
    const float arg = x * fmaf(t, -0.10294324f, -2.3022082f);                             // -2u log2(e)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
    const float t = x * x;
    const float arg = x * fmaf(t, -0.10294324f, -2.3022082f);
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
    const float du2 = 1.5957691216f * fmaf(0.134145f, t, 1.0f);                            // d(2u)/dx = 2 c0 (1 + 3 c1 x^2)
    return fmaf(x * sg * (1.0f - sg), du2, sg);
}

// Pitched forms for the head projection's hidden layer ([M, 4304] inside rows of 4544 fp16 elements):
//   act_rows:      a_bf16[r, c] = act(h[r, c])                  (operand of dW2 = dY^T a; dense [rows, cols])
//   act_bwd_rows:  da[r, c]    *= act'(h[r, c])  in place       (dense [rows, cols] bf16)
// cols % 8 == 0; one 16-byte vector per thread.  TANH: gelu_pytorch_tanh, else erf.
template <bool TANH>
__global__ __launch_bounds__(256) void act_rows_kernel(const _Float16* h, long ldh, long rows, int c8, uint16_t* abf) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8) return;
    const long r = i / c8;
    const int c = (int)(i - r * c8);
    const half8 hv = *reinterpret_cast<const half8*>(h + r * ldh + 8 * c);
    u32x4 bv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = TANH ? gelu_tanh_val((float)hv[2 * e]) : gelu_erf((float)hv[2 * e]);
        const float hi = TANH ? gelu_tanh_val((float)hv[2 * e + 1]) : gelu_erf((float)hv[2 * e + 1]);
        bv[e] = f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
    }
    reinterpret_cast<u32x4*>(abf)[i] = bv;
}
template <bool TANH>
__global__ __launch_bounds__(256) void act_bwd_rows_kernel(uint16_t* da, const _Float16* h, long ldh, long rows, int c8) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8) return;
    const long r = i / c8;
    const int c = (int)(i - r * c8);
    const half8 hv = *reinterpret_cast<const half8*>(h + r * ldh + 8 * c);
    u32x4 d = reinterpret_cast<const u32x4*>(da)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g0 = TANH ? gelu_tanh_grad((float)hv[2 * e]) : gelu_erf_grad((float)hv[2 * e]);
        const float g1 = TANH ? gelu_tanh_grad((float)hv[2 * e + 1]) : gelu_erf_grad((float)hv[2 * e + 1]);
        d[e] = f32_to_bf16(bf16lo_to_f32(d[e]) * g0) | ((uint32_t)f32_to_bf16(bf16hi_to_f32(d[e]) * g1) << 16);
    }
    reinterpret_cast<u32x4*>(da)[i] = d;
}

// a16 = GELU(h) as fp16 (operand of the recomputed second GEMM) and abf = GELU(h) as bf16 (operand of dW2 = dy^T a): n elements, n % 8 == 0
__global__ __launch_bounds__(256) void gelu_split_kernel(const _Float16* h, _Float16* a16, uint16_t* abf, long n8) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const half8 hv = reinterpret_cast<const half8*>(h)[i];
    half8 av;
    u32x4 bv;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { f[e] = gelu_erf((float)hv[e]); av[e] = (_Float16)fminf(fmaxf(f[e], -65504.f), 65504.f); }
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = f32_to_bf16(f[2 * e]) | ((uint32_t)f32_to_bf16(f[2 * e + 1]) << 16);
    if (a16) reinterpret_cast<half8*>(a16)[i] = av;
    reinterpret_cast<u32x4*>(abf)[i] = bv;
}

// da (bf16, in place) *= GELU'(h)  (h fp16 pre-activation): the gradient through the adaptor's hidden activation
__global__ __launch_bounds__(256) void gelu_bwd_kernel(uint16_t* da, const _Float16* h, long n8) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const half8 hv = reinterpret_cast<const half8*>(h)[i];
    u32x4 d = reinterpret_cast<const u32x4*>(da)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = bf16lo_to_f32(d[e]) * gelu_erf_grad((float)hv[2 * e]);
        const float hi = bf16hi_to_f32(d[e]) * gelu_erf_grad((float)hv[2 * e + 1]);
        d[e] = f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
    }
    reinterpret_cast<u32x4*>(da)[i] = d;
}

// the same step with a wave per row and the rows strided over the grid, plus this workgroup's column sums of the result (as stored):
// the bias gradient db1 = sum over the tokens of d h1 without another pass over the 107-MB matrix
template <int NV>
__global__ __launch_bounds__(256) void gelu_bwd_rows_kernel(uint16_t* da, const _Float16* h, long N, float* col_parts) {
    constexpr int D = NV * 384;
    __shared__ float red[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float csum[NV][6];
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) csum[s][j] = 0.f;
    for (long r = (long)blockIdx.x * 4 + wave; r < N; r += (long)gridDim.x * 4) {
        float d[NV][6], hv[NV][6];
        load_row<NV>(da + r * D, lane, d);
        load_row_f16<NV>(h + r * D, lane, hv);
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            float v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = d[s][j] * gelu_erf_grad(hv[s][j]);
            Seg12 w;
            w.a = f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            w.b = f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            w.c = f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
            *reinterpret_cast<Seg12*>(da + r * D + 384 * s + 6 * lane) = w;
            csum[s][0] += bf16lo_to_f32(w.a); csum[s][1] += bf16hi_to_f32(w.a); csum[s][2] += bf16lo_to_f32(w.b);
            csum[s][3] += bf16hi_to_f32(w.b); csum[s][4] += bf16lo_to_f32(w.c); csum[s][5] += bf16hi_to_f32(w.c);
        }
    }
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) red[wave][384 * s + 6 * lane + j] = csum[s][j];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) col_parts[(long)blockIdx.x * D + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// column sums of a bf16 [N, D] matrix (bias gradients: db = sum over the tokens): partial [gridDim.x][D] f32, a wave per row, rows
// strided over the grid; summed by hicom_partials_sum_fwd (block order: deterministic)
template <int NV>
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t* x, long N, float* parts) {
    constexpr int D = NV * 384;
    __shared__ float red[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[NV][6];
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[s][j] = 0.f;
    for (long r = (long)blockIdx.x * 4 + wave; r < N; r += (long)gridDim.x * 4) {
        float v[NV][6];
        load_row<NV>(x + r * D, lane, v);
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[s][j] += v[s][j];
    }
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 6; ++j) red[wave][384 * s + 6 * lane + j] = acc[s][j];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) parts[(long)blockIdx.x * D + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ---- pooled per-window query (trilinear, align_corners=False; projector.py:539-540) ----------
struct PoolParams {
    const uint16_t* x;
    int T, H, W, D, To, Ho, Wo;
    float* out;
};

__device__ __forceinline__ void lerp_tap(int i, int n_in, int n_out, int& i0, int& i1, float& lam) {
    // src = (i + 0.5) * n_in / n_out - 0.5 clamped at 0 (PyTorch area_pixel_compute_source_index)
    const float scale = (float)n_in / (float)n_out;
    float src = ((float)i + 0.5f) * scale - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)floorf(src);
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    lam = src - (float)i0;
}

__global__ __launch_bounds__(256) void trilinear_pool_kernel(PoolParams p) {
    const int o = blockIdx.x;
    const int wo = o % p.Wo, ho = (o / p.Wo) % p.Ho, to = o / (p.Wo * p.Ho);
    int t0, t1, y0, y1, x0, x1;
    float lt, ly, lx;
    lerp_tap(to, p.T, p.To, t0, t1, lt);
    lerp_tap(ho, p.H, p.Ho, y0, y1, ly);
    lerp_tap(wo, p.W, p.Wo, x0, x1, lx);
    auto at = [&](int t, int y, int x, int c) -> float {
        return bf16_to_f32(p.x[(((long)t * p.H + y) * p.W + x) * p.D + c]);
    };
    for (int c = threadIdx.x; c < p.D; c += blockDim.x) {
        // same nesting as the separable restatement in the oracle: t, then h, then w
        const float a00 = at(t0, y0, x0, c) * (1.f - lt) + at(t1, y0, x0, c) * lt;
        const float a01 = at(t0, y0, x1, c) * (1.f - lt) + at(t1, y0, x1, c) * lt;
        const float a10 = at(t0, y1, x0, c) * (1.f - lt) + at(t1, y1, x0, c) * lt;
        const float a11 = at(t0, y1, x1, c) * (1.f - lt) + at(t1, y1, x1, c) * lt;
        const float b0 = a00 * (1.f - ly) + a10 * ly;
        const float b1 = a01 * (1.f - ly) + a11 * ly;
        p.out[(long)o * p.D + c] = b0 * (1.f - lx) + b1 * lx;
    }
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_local_attn_fwd(const void* key, int32_t key_dt, const void* value, int32_t value_dt, int32_t D,
                                    hicom_axis at, hicom_axis ay, hicom_axis ax,
                                    const void* query, int32_t query_dt, int64_t query_stride,
                                    float scale, float bias, int32_t l2norm,
                                    float* ctx, void* ctx_f16, void* stream) {
    HICOM_REQUIRE(key && value && query && (ctx || ctx_f16), HICOM_EINVAL, "local_attn: NULL pointer");
    HICOM_REQUIRE(D == 1152 || D == 768, HICOM_EUNSUP, "local_attn: D=%d (only 1152 / 768)", D);
    HICOM_REQUIRE(query_dt == HICOM_DT_BF16 || query_dt == HICOM_DT_F32, HICOM_EINVAL, "local_attn: query dtype");
    HICOM_REQUIRE(key_dt >= 0 && key_dt <= 2 && value_dt >= 0 && value_dt <= 2, HICOM_EINVAL, "local_attn: stream dtype");
    for (const hicom_axis* a : {&at, &ay, &ax}) {
        HICOM_REQUIRE(a->n > 0 && a->k > 0 && a->nwin > 0 && a->nfull >= 0 && a->nfull <= a->nwin && a->k <= a->n,
                      HICOM_EINVAL, "local_attn: bad axis n=%d k=%d nwin=%d nfull=%d", a->n, a->k, a->nwin, a->nfull);
        const int last = axis_start(*a, a->nwin - 1);
        HICOM_REQUIRE(last >= 0 && last + a->k <= a->n, HICOM_EINVAL, "local_attn: window runs off the axis");
    }
    const long win = (long)at.k * ay.k * ax.k;
    HICOM_REQUIRE(win <= 4096, HICOM_EUNSUP, "local_attn: window of %ld tokens is too large", win);
    const long nwin = (long)at.nwin * ay.nwin * ax.nwin;
    HICOM_REQUIRE(nwin < (1L << 31), HICOM_EINVAL, "local_attn: too many windows");
    LocalParams p{key, value, key_dt, value_dt, query, query_dt == HICOM_DT_F32,
                  (long)query_stride, at, ay, ax, scale, bias, l2norm, ctx, (_Float16*)ctx_f16};
    const size_t smem = (((size_t)win + 3) & ~(size_t)3) * 4 + 4 * (size_t)D * 4;
    hipStream_t s = (hipStream_t)stream;
    if (D == 1152) hipLaunchKernelGGL(local_attn_kernel<3>, dim3((unsigned)nwin), dim3(256), smem, s, p);
    else hipLaunchKernelGGL(local_attn_kernel<2>, dim3((unsigned)nwin), dim3(256), smem, s, p);
    return hicom_host::check_launch("local_attn");
}

extern "C" int hicom_local_attn_adapt_fwd(const void* key_x, const void* key_y, const void* k_gamma, const void* k_beta, const void* k_alpha,
                                          const void* value_x, const void* value_y, const void* v_gamma, const void* v_beta, const void* v_alpha,
                                          int32_t alpha_dt, float eps, int32_t D, hicom_axis at, hicom_axis ay, hicom_axis ax,
                                          const void* query, int32_t query_dt, int64_t query_stride, float scale, float bias,
                                          float* ctx, void* stream) {
    HICOM_REQUIRE(key_x && value_x && query && ctx && (key_y || value_y), HICOM_EINVAL, "local_attn_adapt: NULL pointer / no adapted stream");
    HICOM_REQUIRE(!key_y || (k_gamma && k_beta && k_alpha), HICOM_EINVAL, "local_attn_adapt: key adaptor parameters");
    HICOM_REQUIRE(!value_y || (v_gamma && v_beta && v_alpha), HICOM_EINVAL, "local_attn_adapt: value adaptor parameters");
    HICOM_REQUIRE(D == 1152 || D == 768, HICOM_EUNSUP, "local_attn_adapt: D=%d (only 1152 / 768)", D);
    HICOM_REQUIRE(query_dt == HICOM_DT_BF16 || query_dt == HICOM_DT_F32, HICOM_EINVAL, "local_attn_adapt: query dtype");
    HICOM_REQUIRE(alpha_dt == HICOM_DT_BF16 || alpha_dt == HICOM_DT_F32, HICOM_EINVAL, "local_attn_adapt: alpha dtype");
    for (const hicom_axis* a : {&at, &ay, &ax}) {
        HICOM_REQUIRE(a->n > 0 && a->k > 0 && a->nwin > 0 && a->nfull >= 0 && a->nfull <= a->nwin && a->k <= a->n,
                      HICOM_EINVAL, "local_attn_adapt: bad axis n=%d k=%d nwin=%d nfull=%d", a->n, a->k, a->nwin, a->nfull);
        const int last = axis_start(*a, a->nwin - 1);
        HICOM_REQUIRE(last >= 0 && last + a->k <= a->n, HICOM_EINVAL, "local_attn_adapt: window runs off the axis");
    }
    const long win = (long)at.k * ay.k * ax.k;
    HICOM_REQUIRE(win <= 4096, HICOM_EUNSUP, "local_attn_adapt: window of %ld tokens is too large", win);
    const long nwin = (long)at.nwin * ay.nwin * ax.nwin;
    HICOM_REQUIRE(nwin < (1L << 31), HICOM_EINVAL, "local_attn_adapt: too many windows");
    LocalAdaptParams p{(const uint16_t*)key_x, (const _Float16*)key_y, (const uint16_t*)k_gamma, (const uint16_t*)k_beta, k_alpha,
                       (const uint16_t*)value_x, (const _Float16*)value_y, (const uint16_t*)v_gamma, (const uint16_t*)v_beta, v_alpha,
                       alpha_dt == HICOM_DT_F32, eps, query, query_dt == HICOM_DT_F32, (long)query_stride, at, ay, ax, scale, bias, ctx};
    const size_t smem = (((size_t)win + 3) & ~(size_t)3) * 4 + 4 * (size_t)D * 4;
    hipStream_t s = (hipStream_t)stream;
    if (D == 1152) hipLaunchKernelGGL(local_attn_adapt_kernel<3>, dim3((unsigned)nwin), dim3(256), smem, s, p);
    else hipLaunchKernelGGL(local_attn_adapt_kernel<2>, dim3((unsigned)nwin), dim3(256), smem, s, p);
    return hicom_host::check_launch("local_attn_adapt");
}

extern "C" int hicom_local_attn_bwd(const void* key, const void* value, int32_t D,
                                    hicom_axis at, hicom_axis ay, hicom_axis ax,
                                    const void* query, int32_t query_dt, int64_t query_stride,
                                    float scale, float bias, const float* dctx, float* dq, void* dkey,
                                    int32_t l2norm_key, float* dls, void* dvalue, int32_t value_is_key, void* stream) {
    HICOM_REQUIRE(key && value && query && dctx && dq, HICOM_EINVAL, "local_attn_bwd: NULL pointer");
    HICOM_REQUIRE(D == 1152 || D == 768, HICOM_EUNSUP, "local_attn_bwd: D=%d (only 1152 / 768)", D);
    HICOM_REQUIRE(query_dt == HICOM_DT_BF16 || query_dt == HICOM_DT_F32, HICOM_EINVAL, "local_attn_bwd: query dtype");
    for (const hicom_axis* a : {&at, &ay, &ax}) {
        HICOM_REQUIRE(a->n > 0 && a->k > 0 && a->nwin > 0 && a->nfull >= 0 && a->nfull <= a->nwin && a->k <= a->n,
                      HICOM_EINVAL, "local_attn_bwd: bad axis n=%d k=%d nwin=%d nfull=%d", a->n, a->k, a->nwin, a->nfull);
        const int last = axis_start(*a, a->nwin - 1);
        HICOM_REQUIRE(last >= 0 && last + a->k <= a->n, HICOM_EINVAL, "local_attn_bwd: window runs off the axis");
    }
    const long win = (long)at.k * ay.k * ax.k;
    HICOM_REQUIRE(win <= 4096, HICOM_EUNSUP, "local_attn_bwd: window of %ld tokens is too large", win);
    const long nwin = (long)at.nwin * ay.nwin * ax.nwin;
    HICOM_REQUIRE(nwin < (1L << 31), HICOM_EINVAL, "local_attn_bwd: too many windows");
    LocalBwdParams p{(const uint16_t*)key, (const uint16_t*)value, query, query_dt == HICOM_DT_F32, (long)query_stride, at, ay, ax,
                     scale, bias, dctx, dq, (uint16_t*)dkey, l2norm_key ? 1 : 0, dls, (uint16_t*)dvalue, value_is_key ? 1 : 0, 0, 0, 0};
    HICOM_REQUIRE(!(value_is_key && dkey), HICOM_EINVAL, "local_attn_bwd: value_is_key writes the summed gradient to dvalue (dkey must be NULL)");
    const size_t smem = 5 * (((size_t)win + 3) & ~(size_t)3) * 4 + 4 * (size_t)D * 4;
    hipStream_t s = (hipStream_t)stream;
    // per-token outputs over OVERLAPPING windows: cleared, then one accumulating launch per parity class of the axes that overlap
    int mask = 0;
    if (dkey || dvalue) {
        const hicom_axis* axs[3] = {&at, &ay, &ax};
        for (int a = 0; a < 3; ++a)
            if ((long)axs[a]->nwin * axs[a]->k != axs[a]->n) mask |= 1 << a;
    }
    if (mask) {
        const size_t bytes = (size_t)at.n * ay.n * ax.n * D * 2;
        if (dkey) HICOM_REQUIRE(hipMemsetAsync(dkey, 0, bytes, s) == hipSuccess, HICOM_ELAUNCH, "local_attn_bwd: memset");
        if (dvalue) HICOM_REQUIRE(hipMemsetAsync(dvalue, 0, bytes, s) == hipSuccess, HICOM_ELAUNCH, "local_attn_bwd: memset");
        p.accumulate = 1;
        p.par_mask = mask;
    }
    for (int val = 0; val < 8; ++val) {
        if (val & ~mask) continue;
        p.par_val = val;
        if (D == 1152) hipLaunchKernelGGL(local_attn_bwd_kernel<3>, dim3((unsigned)nwin), dim3(256), smem, s, p);
        else hipLaunchKernelGGL(local_attn_bwd_kernel<2>, dim3((unsigned)nwin), dim3(256), smem, s, p);
    }
    return hicom_host::check_launch("local_attn_bwd");
}

static int check_axes(const char* who, hicom_axis at, hicom_axis ay, hicom_axis ax, bool exact) {
    for (const hicom_axis* a : {&at, &ay, &ax}) {
        HICOM_REQUIRE(a->n > 0 && a->k > 0 && a->nwin > 0 && a->nfull >= 0 && a->nfull <= a->nwin && a->k <= a->n,
                      HICOM_EINVAL, "%s: bad axis n=%d k=%d nwin=%d nfull=%d", who, a->n, a->k, a->nwin, a->nfull);
        if (exact) HICOM_REQUIRE((long)a->nwin * a->k == a->n, HICOM_EUNSUP, "%s: needs an exact window partition (n=%d k=%d)", who, a->n, a->k);
    }
    return HICOM_OK;
}

extern "C" int hicom_local_attn_adapt_bwd(const void* key_x, const void* key_y, const void* k_gamma, const void* k_beta, const void* k_alpha,
                                          const void* value_x, const void* value_y, const void* v_gamma, const void* v_beta, const void* v_alpha,
                                          int32_t alpha_dt, float eps, int32_t D, hicom_axis at, hicom_axis ay, hicom_axis ax,
                                          const void* query, int32_t query_dt, int64_t query_stride, float scale, float bias,
                                          const float* dctx, float* ds, float* pw, float* sxk, float* syk, float* sxv, float* syv, void* stream) {
    HICOM_REQUIRE(key_x && value_x && query && dctx && ds && pw && sxk && sxv && (key_y || value_y), HICOM_EINVAL, "local_attn_adapt_bwd: NULL pointer");
    HICOM_REQUIRE((!key_y || (k_gamma && k_beta && k_alpha && syk)) && (!value_y || (v_gamma && v_beta && v_alpha && syv)), HICOM_EINVAL,
                  "local_attn_adapt_bwd: adaptor parameters / outputs");
    HICOM_REQUIRE(D == 1152 || D == 768, HICOM_EUNSUP, "local_attn_adapt_bwd: D=%d (only 1152 / 768)", D);
    HICOM_REQUIRE((query_dt == HICOM_DT_BF16 || query_dt == HICOM_DT_F32) && (alpha_dt == HICOM_DT_BF16 || alpha_dt == HICOM_DT_F32), HICOM_EINVAL,
                  "local_attn_adapt_bwd: query / alpha dtype");
    if (int rc = check_axes("local_attn_adapt_bwd", at, ay, ax, true)) return rc;      // (ds / pw are written once per token)
    const long win = (long)at.k * ay.k * ax.k, nwin = (long)at.nwin * ay.nwin * ax.nwin;
    HICOM_REQUIRE(win <= 1024 && nwin < (1L << 31), HICOM_EUNSUP, "local_attn_adapt_bwd: window of %ld tokens is too large", win);
    LocalAdaptBwdParams p{(const uint16_t*)key_x, (const _Float16*)key_y, (const uint16_t*)k_gamma, (const uint16_t*)k_beta, k_alpha,
                          (const uint16_t*)value_x, (const _Float16*)value_y, (const uint16_t*)v_gamma, (const uint16_t*)v_beta, v_alpha,
                          alpha_dt == HICOM_DT_F32, eps, query, query_dt == HICOM_DT_F32, (long)query_stride, at, ay, ax, scale, bias,
                          dctx, ds, pw, sxk, syk, sxv, syv};
    const size_t smem = 6 * (((size_t)win + 3) & ~(size_t)3) * 4 + 4 * (size_t)D * 4;
    hipStream_t s = (hipStream_t)stream;
    if (D == 1152) hipLaunchKernelGGL(local_attn_adapt_bwd_kernel<3>, dim3((unsigned)nwin), dim3(256), smem, s, p);
    else hipLaunchKernelGGL(local_attn_adapt_bwd_kernel<2>, dim3((unsigned)nwin), dim3(256), smem, s, p);
    return hicom_host::check_launch("local_attn_adapt_bwd");
}

extern "C" int hicom_adapt_dy_fwd(const void* y, const void* gamma, const void* vec, int32_t vec_dt, int64_t vec_stride, const float* coef,
                                  const void* alpha, int32_t alpha_dt, float eps, int32_t D, hicom_axis at, hicom_axis ay, hicom_axis ax,
                                  void* dy, void* r1, float* col_parts, int32_t nparts, void* stream) {
    HICOM_REQUIRE(y && gamma && vec && coef && alpha && dy && (!col_parts || nparts > 0), HICOM_EINVAL, "adapt_dy: NULL pointer");
    HICOM_REQUIRE(D == 1152 || D == 768, HICOM_EUNSUP, "adapt_dy: D=%d (only 1152 / 768)", D);
    HICOM_REQUIRE((vec_dt == HICOM_DT_BF16 || vec_dt == HICOM_DT_F32) && (alpha_dt == HICOM_DT_BF16 || alpha_dt == HICOM_DT_F32), HICOM_EINVAL, "adapt_dy: dtypes");
    if (int rc = check_axes("adapt_dy", at, ay, ax, true)) return rc;
    const long N = (long)at.n * ay.n * ax.n;
    AdaptDyParams p{(const _Float16*)y, (const uint16_t*)gamma, vec, vec_dt == HICOM_DT_F32, (long)vec_stride, coef, alpha, alpha_dt == HICOM_DT_F32, eps,
                    at, ay, ax, (uint16_t*)dy, (uint16_t*)r1, N, col_parts};
    hipStream_t s = (hipStream_t)stream;
    // with column partials: `nparts` workgroups walk the tokens (each leaves one partial row); without: one token per wave
    const unsigned grid = col_parts ? (unsigned)nparts : (unsigned)((N + 3) / 4);
    if (D == 1152) hipLaunchKernelGGL(adapt_dy_kernel<3>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(adapt_dy_kernel<2>, dim3(grid), dim3(256), 0, s, p);
    return hicom_host::check_launch("adapt_dy");
}

extern "C" int hicom_gelu_split_fwd(const void* h_f16, void* a_f16, void* a_bf16, int64_t n, void* stream) {
    HICOM_REQUIRE(h_f16 && a_bf16 && n > 0 && n % 8 == 0 && ((uintptr_t)h_f16 % 16 == 0) && ((uintptr_t)a_f16 % 16 == 0) && ((uintptr_t)a_bf16 % 16 == 0),
                  HICOM_EINVAL, "gelu_split: bad arguments (n %% 8, 16-byte alignment)");
    hipLaunchKernelGGL(gelu_split_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)h_f16, (_Float16*)a_f16,
                       (uint16_t*)a_bf16, (long)(n / 8));
    return hicom_host::check_launch("gelu_split");
}

extern "C" int hicom_gelu_bwd_fwd(void* da_bf16, const void* h_f16, int64_t n, int32_t D, float* col_parts, int32_t nparts, void* stream) {
    HICOM_REQUIRE(da_bf16 && h_f16 && n > 0 && n % 8 == 0 && ((uintptr_t)da_bf16 % 16 == 0) && ((uintptr_t)h_f16 % 16 == 0), HICOM_EINVAL,
                  "gelu_bwd: bad arguments (n %% 8, 16-byte alignment)");
    if (col_parts) {
        HICOM_REQUIRE((D == 1152 || D == 768) && n % D == 0 && nparts > 0, HICOM_EINVAL, "gelu_bwd: column partials need rows of D = 1152 / 768 and nparts > 0");
        if (D == 1152) hipLaunchKernelGGL(gelu_bwd_rows_kernel<3>, dim3((unsigned)nparts), dim3(256), 0, (hipStream_t)stream, (uint16_t*)da_bf16, (const _Float16*)h_f16, (long)(n / D), col_parts);
        else hipLaunchKernelGGL(gelu_bwd_rows_kernel<2>, dim3((unsigned)nparts), dim3(256), 0, (hipStream_t)stream, (uint16_t*)da_bf16, (const _Float16*)h_f16, (long)(n / D), col_parts);
        return hicom_host::check_launch("gelu_bwd");
    }
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)da_bf16, (const _Float16*)h_f16,
                       (long)(n / 8));
    return hicom_host::check_launch("gelu_bwd");
}

extern "C" int hicom_act_rows_fwd(const void* h_f16, int64_t ldh, int64_t rows, int32_t cols, int32_t act, void* a_bf16, void* stream) {
    HICOM_REQUIRE(h_f16 && a_bf16 && rows > 0 && cols > 0 && cols % 8 == 0 && ldh >= cols && ldh % 8 == 0 && ((uintptr_t)h_f16 % 16 == 0) &&
                      ((uintptr_t)a_bf16 % 16 == 0) && (act == HICOM_ACT_GELU || act == 2), HICOM_EINVAL,
                  "act_rows: bad arguments (cols, ldh %% 8; 16-byte alignment; act GELU | GELU_TANH)");
    const long n = rows * (cols / 8);
    if (act == 2) hipLaunchKernelGGL(act_rows_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)h_f16, (long)ldh, (long)rows, cols / 8, (uint16_t*)a_bf16);
    else hipLaunchKernelGGL(act_rows_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)h_f16, (long)ldh, (long)rows, cols / 8, (uint16_t*)a_bf16);
    return hicom_host::check_launch("act_rows");
}

extern "C" int hicom_act_bwd_rows_fwd(void* da_bf16, const void* h_f16, int64_t ldh, int64_t rows, int32_t cols, int32_t act, void* stream) {
    HICOM_REQUIRE(h_f16 && da_bf16 && rows > 0 && cols > 0 && cols % 8 == 0 && ldh >= cols && ldh % 8 == 0 && ((uintptr_t)h_f16 % 16 == 0) &&
                      ((uintptr_t)da_bf16 % 16 == 0) && (act == HICOM_ACT_GELU || act == 2), HICOM_EINVAL,
                  "act_bwd_rows: bad arguments (cols, ldh %% 8; 16-byte alignment; act GELU | GELU_TANH)");
    const long n = rows * (cols / 8);
    if (act == 2) hipLaunchKernelGGL(act_bwd_rows_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)da_bf16, (const _Float16*)h_f16, (long)ldh, (long)rows, cols / 8);
    else hipLaunchKernelGGL(act_bwd_rows_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)da_bf16, (const _Float16*)h_f16, (long)ldh, (long)rows, cols / 8);
    return hicom_host::check_launch("act_bwd_rows");
}

extern "C" int hicom_colsum_fwd(const void* x_bf16, int64_t N, int32_t D, float* parts, int32_t nparts, void* stream) {
    HICOM_REQUIRE(x_bf16 && parts && N > 0 && nparts > 0 && (D == 1152 || D == 768), HICOM_EINVAL, "colsum: bad arguments (D 1152 / 768)");
    if (D == 1152) hipLaunchKernelGGL(colsum_kernel<3>, dim3((unsigned)nparts), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16, (long)N, parts);
    else hipLaunchKernelGGL(colsum_kernel<2>, dim3((unsigned)nparts), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16, (long)N, parts);
    return hicom_host::check_launch("colsum");
}

extern "C" int hicom_trilinear_pool_fwd(const void* x, int32_t T, int32_t H, int32_t W, int32_t D,
                                        int32_t To, int32_t Ho, int32_t Wo, float* out, void* stream) {
    HICOM_REQUIRE(x && out, HICOM_EINVAL, "trilinear_pool: NULL pointer");
    HICOM_REQUIRE(T > 0 && H > 0 && W > 0 && D > 0 && To > 0 && Ho > 0 && Wo > 0, HICOM_EINVAL, "trilinear_pool: bad shape");
    PoolParams p{(const uint16_t*)x, T, H, W, D, To, Ho, Wo, out};
    hipLaunchKernelGGL(trilinear_pool_kernel, dim3((unsigned)((long)To * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, p);
    return hicom_host::check_launch("trilinear_pool");
}
