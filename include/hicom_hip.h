/*
 * hicom_hip.h -- C ABI of libhicom_hip.so: MI355X (gfx950 / CDNA4) kernels for the HICom
 * hybrid-level video-token compressor.
 *
 * The reference (lntzm/HICom) is pure Python and has no FFI; each entry point below replaces
 * a chain of stock PyTorch ops on the path
 *     hicom/model/projector.py:676-708   HIComProjector.forward
 * and is cited to the reference lines it stands in for.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (torch on the Python
 *     side); the library allocates nothing and keeps no state between calls;
 *   - dense row-major layouts, no strides; "bf16" = uint16_t bit patterns; f32 = float;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued asynchronously on it and nothing synchronises the host (graph-capturable);
 *   - return 0 on success, a negative HICOM_E* code otherwise; never throws.
 *     hicom_last_error() returns a thread-local description of the last failure.
 */
#ifndef HICOM_HIP_H
#define HICOM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HICOM_ABI_VERSION 15

#define HICOM_OK         0
#define HICOM_EINVAL    -1   /* bad argument (shape, alignment, NULL)        */
#define HICOM_EUNSUP    -2   /* configuration not supported by the kernels   */
#define HICOM_ELAUNCH   -3   /* HIP launch / runtime error                   */

#define HICOM_DT_BF16 0
#define HICOM_DT_F32  1
#define HICOM_DT_F16  2      /* IEEE half: activations that feed matrix cores (11 significand bits), accepted where noted */

/* activation / epilogue flags for the linear + GEMM entry points */
#define HICOM_ACT_NONE 0
#define HICOM_ACT_GELU 1     /* exact erf GELU (nn.GELU() default, projector.py:310) */
#define HICOM_ACT_GELU_TANH 2 /* tanh GELU (HF "gelu_pytorch_tanh": the SigLIP MLP, encoder.py:285) */

int hicom_abi_version(void);
const char* hicom_last_error(void);

/* One axis of the window tiling of LocalCompressor.divide_feature /
 * balance_divide_feature (projector.py:473-522): `nwin` windows of length `k` over `n`
 * elements; the first `nfull` start at i*k, the rest at nfull*k + (i-nfull)*(k-1) - 1
 * (one-element overlap).  Divisible axes have nfull == nwin. */
typedef struct hicom_axis {
    int32_t n, k, nwin, nfull;
} hicom_axis;

/* ---- local compressor: windowed single-head cross-attention ------------------------------
 * Replaces projector.py:544-558 (divide_feature x3, bmm, softmax, bmm, un-window) and, with
 * l2norm != 0, the clip-scale normalisation at :527-529,549.
 *   key, value : [T,H,W,D] bf16 (frames_embed / frames_feature; key may alias value), or f32 / fp16 (the
 *                alpha-blended adaptor outputs of projector.py:533-534); dtype key_dt / value_dt
 *   query      : [D] shared by every window (query_stride == 0, GuideInjector "direct",
 *                projector.py:352-368) or [Nw, D] with row stride query_stride elements;
 *                dtype query_dt
 *   logits     : (q.k) * scale + bias        (scale = 1/sqrt(D) or exp(logit_scale))
 *   ctx        : f32 [Nw, D], window order (t1,h1,w1) row-major, Nw = at.nwin*ay.nwin*ax.nwin; and / or
 *   ctx_f16    : the same contexts as one fp16 plane (saturating): the operand of hicom_readout16_gemm_fwd (either may be NULL)
 *   l2norm     : bit 0 = L2-normalise every key row, bit 1 = L2-normalise the query
 *                (clip-scale variant, projector.py:527-529)
 * D must be 1152 or 768 (projector.py:407-414). */
int hicom_local_attn_fwd(const void* key, int32_t key_dt, const void* value, int32_t value_dt, int32_t D,
                         hicom_axis at, hicom_axis ay, hicom_axis ax,
                         const void* query, int32_t query_dt, int64_t query_stride,
                         float scale, float bias, int32_t l2norm,
                         float* ctx, void* ctx_f16, void* stream);

/* ---- windowed attention with the k / v adaptor blend fused into its row loads (projector.py:533-534 in front of :544-553) ----
 *   key_n = (1 - a_k) key_x_n + a_k (LayerNorm_eps(key_y_n) k_gamma + k_beta),  key_y = k_proj(key_x) (the two dense GEMMs of the
 *   adaptor MLP, fp16 [N, D]); likewise the values.  The blended streams are never materialised: logits and contexts are formed
 *   from the x and y rows with the LayerNorm statistics computed on the row in registers.  key_y or value_y may be NULL (that stream
 *   has no adaptor).  key_x / value_x bf16 [T,H,W,D]; gamma / beta bf16 [D]; alpha: device scalar (bf16 | f32).  No clip-scale. */
int hicom_local_attn_adapt_fwd(const void* key_x, const void* key_y, const void* k_gamma, const void* k_beta, const void* k_alpha,
                               const void* value_x, const void* value_y, const void* v_gamma, const void* v_beta, const void* v_alpha,
                               int32_t alpha_dt, float eps, int32_t D, hicom_axis at, hicom_axis ay, hicom_axis ax,
                               const void* query, int32_t query_dt, int64_t query_stride, float scale, float bias,
                               float* ctx, void* stream);

/* ---- backward of the windowed attention with the k / v adaptor blends (training path of `local43_adaptkv_global32`; autograd
 * through projector.py:533-534 + :550-553).  Arguments as hicom_local_attn_adapt_fwd plus dctx f32 [Nw, D].  Outputs (exact window
 * partition required):
 *   ds [N] f32 = scale dS_n,  pw [N] f32 = p_n       token-indexed: d key_n = ds_n q_w, d value_n = pw_n dctx_w (rank 1, never written)
 *   sxk, sxv [Nw, D] f32 = sum_n ds_n key_x_n, sum_n pw_n value_x_n;   syk, syv [Nw, D] = sum_n ds_n yhat_k,n, sum_n pw_n yhat_v,n with
 *   yhat = LayerNorm-normalised y (no affine); syk / syv may be NULL when that stream has no adaptor.
 * From these: dq_w = (1 - a_k) sxk_w + a_k gamma_k syk_w, and d alpha / d gamma / d beta of both adaptors as sums over the windows.
 * hicom_adapt_dy_fwd continues into the adaptor MLP: dy[tok] = d/dy of alpha LN(y) gamma for the upstream gradient coef[tok] * vec[w(tok)]
 * (K: coef = ds, vec = q; V: coef = pw, vec = dctx), written as bf16 [N, D]; r1 (may be NULL) = (1 - alpha) coef[tok] vec[w] as bf16, the
 * x-branch of d frames_embed.  hicom_gelu_split_fwd / hicom_gelu_bwd_fwd / hicom_colsum_fwd: the elementwise steps of the adaptor-MLP backward
 * (a = GELU(h) as fp16 and bf16; da *= GELU'(h) in place; partial column sums [nparts][D] for the bias gradients).
 * hicom_adapt_dy_fwd and hicom_gelu_bwd_fwd leave the column partials of what they write themselves when col_parts (f32 [nparts][D],
 * summed by hicom_partials_sum_fwd) is given: `nparts` workgroups then walk the rows -- no separate pass for the bias gradients. */
int hicom_local_attn_adapt_bwd(const void* key_x, const void* key_y, const void* k_gamma, const void* k_beta, const void* k_alpha,
                               const void* value_x, const void* value_y, const void* v_gamma, const void* v_beta, const void* v_alpha,
                               int32_t alpha_dt, float eps, int32_t D, hicom_axis at, hicom_axis ay, hicom_axis ax,
                               const void* query, int32_t query_dt, int64_t query_stride, float scale, float bias,
                               const float* dctx, float* ds, float* pw, float* sxk, float* syk, float* sxv, float* syv, void* stream);
int hicom_adapt_dy_fwd(const void* y, const void* gamma, const void* vec, int32_t vec_dt, int64_t vec_stride, const float* coef,
                       const void* alpha, int32_t alpha_dt, float eps, int32_t D, hicom_axis at, hicom_axis ay, hicom_axis ax,
                       void* dy, void* r1, float* col_parts, int32_t nparts, void* stream);
int hicom_gelu_split_fwd(const void* h_f16, void* a_f16 /* may be NULL */, void* a_bf16, int64_t n, void* stream);
/* The same two steps for a hidden layer that lies inside pitched rows (the SigLIP head's [M, 4304] inside rows of 4544 fp16 elements,
 * encoder.py:284-286) and for either activation (act = HICOM_ACT_GELU | 2 = gelu_pytorch_tanh): a_bf16 [rows, cols] = act(h[r, c]);
 * da_bf16 [rows, cols] *= act'(h[r, c]) in place.  cols, ldh multiples of 8. */
int hicom_act_rows_fwd(const void* h_f16, int64_t ldh, int64_t rows, int32_t cols, int32_t act, void* a_bf16, void* stream);
int hicom_act_bwd_rows_fwd(void* da_bf16, const void* h_f16, int64_t ldh, int64_t rows, int32_t cols, int32_t act, void* stream);
int hicom_gelu_bwd_fwd(void* da_bf16, const void* h_f16, int64_t n, int32_t D, float* col_parts, int32_t nparts, void* stream);
int hicom_colsum_fwd(const void* x_bf16, int64_t N, int32_t D, float* parts, int32_t nparts, void* stream);

/* ---- backward of the windowed attention (training path; autograd through projector.py:550-553) ----------
 * For the reference's stage 3 (train.py:717-726: the SigLIP head and the guide encoder train too) the gradients
 * w.r.t. the key stream frames_embed and the query are needed.
 *   key, value : bf16 [T,H,W,D];  query as in hicom_local_attn_fwd;  dctx : f32 [Nw, D] upstream gradient of ctx
 *   dq         : f32 [Nw, D]  scale * sum_i dS_i k_i per window (a shared query's gradient is the sum over windows)
 *   dkey       : bf16 [T,H,W,D] scale * dS_i * q, or NULL.  Where the kernel does not divide an axis the trailing windows overlap by one plane
 *                (projector.py:501-522): dkey / dvalue are then cleared and accumulated by one launch per parity class of the window index along
 *                every such axis (windows of one class share no token: ordered sums, no atomics)
 *   l2norm_key (ABI 15): clip-scale on the local stage (projector.py:527-529, :549; trainable under `attn_scale`, train.py:730-733):
 *                the key rows enter L2-normalised, s_i = scale (q . k_i / ||k_i||) + bias with scale = e^logit_scale, bias = logit_bias;
 *                dq and dkey then go through the normalisation.  The query is used as given (the caller normalises the guide).
 *   dls        : f32 [Nw] or NULL: sum_i dS_i (s_i - bias) per window = this window's share of d logit_scale (d logit_bias is 0: the
 *                softmax cancels a shift) */
int hicom_local_attn_bwd(const void* key, const void* value, int32_t D,
                         hicom_axis at, hicom_axis ay, hicom_axis ax,
                         const void* query, int32_t query_dt, int64_t query_stride,
                         float scale, float bias, const float* dctx, float* dq, void* dkey,
                         int32_t l2norm_key, float* dls, void* dvalue, int32_t value_is_key, void* stream);
/* (dvalue, ABI 15: bf16 [T,H,W,D] = p_i dctx_w, the gradient w.r.t. the VALUE stream frames_feature, or NULL; value_is_key: the keys are
 * the value rows -- frames_embed None, projector.py:532 -- and the key-side gradient is added into dvalue, dkey NULL.)
 *
 * d frames_feature of the GLOBAL stage, direct recipe (<= 16 folded rows), from the tensors the attention backward leaves:
 *   dx[n, :] (+)= sum_r dS[r, n] qt[r, :] + exp(S[r, n] - M_r) / L_r dctx[r, :]
 * S = the forward's logits, dS = hicom_global_stream_bwd's ds_out (both f32 [rows_pad, score_stride]), ml f32 [rows][2], qt / dctx f32
 * [rows, E] (folded queries incl. the attention scale; upstream gradients of the per-head contexts), dx bf16 [N, E]; accumulate != 0:
 * dx already holds the local stage's share (read, added in fp32, written back).  The positional terms do not depend on x. */
int hicom_global_dx_fwd(const float* S, const float* dS, int64_t score_stride, const float* ml, const float* qt, const float* dctx,
                        int32_t rows, int64_t N, int32_t E, void* dx, int32_t accumulate, void* stream);

/* ---- pooled per-window query: F.interpolate(..., 'trilinear', align_corners=False) --------
 * Replaces projector.py:539-540.  x bf16 [T,H,W,D] -> out f32 [t',h',w',D]. */
int hicom_trilinear_pool_fwd(const void* x, int32_t T, int32_t H, int32_t W, int32_t D,
                             int32_t To, int32_t Ho, int32_t Wo, float* out, void* stream);

/* ---- small-M linear: y[M,N] = act(x[M,K] . w[N,K]^T + b) + res ---------------------------
 * nn.Linear on a handful of rows (q_proj / out_proj / readout of the global tokens,
 * projector.py:180,226,646; also the score-side positional table q~.PE^T).
 *   head_dim > 0 selects "per-head rows": column n reads x row m*head_rows + n/head_dim
 *   (v_proj applied to the per-head context, projector.py:182,215 after folding).
 *   b, res may be NULL.  res is [M,N]; res_flags bit 0 = broadcast row 0, bit 1 = res is bf16
 *   (otherwise f32). */
int hicom_linear_fwd(const void* x, int32_t x_dt, const void* w, int32_t w_dt,
                     const void* b, int32_t b_dt, const void* res, int32_t res_flags,
                     int32_t M, int32_t N, int32_t K, int32_t head_rows, int32_t head_dim,
                     int32_t act, float* y, void* stream);

/* Same product, written straight into a packed output: row m of y goes to rows
 * row0 + m + k*M, k < n_rows / M, of dst [*, ldd] (dtype dst_dt) -- the last Linear of the global
 * readout filling the 32 global rows (projector.py:646,707; in "direct" mode M == 1 and the 32 rows
 * are 32 copies). */
int hicom_linear_to_rows_fwd(const void* x, int32_t x_dt, const void* w, int32_t w_dt,
                             const void* b, int32_t b_dt, int32_t M, int32_t N, int32_t K, int32_t act,
                             void* dst, int32_t dst_dt, int64_t ldd, int64_t row0, int32_t n_rows,
                             void* stream);

/* ---- fold k_proj into the queries ---------------------------------------------------------
 * qt[(q*nh + h), c] = scale * sum_j w_k[h*hd + j, c] * qp[q, h*hd + j]
 * so that score_h(n) = qt_h . x_n (+ a key-independent constant that softmax cancels):
 * replaces k_proj over all T*729 tokens (projector.py:181,193-197).  qp f32 [nq,E],
 * w_k bf16 [E,E] (nn.Linear layout [out,in]), qt f32 [nq*nh, E]. */
int hicom_fold_query_fwd(const float* qp, const void* w_k, int32_t nq, int32_t nh, int32_t E,
                         float scale, float* qt, void* stream);

/* Fused form used on the hot path: emits qt directly as the bf16 hi/lo pair of the stream kernel
 * and, when kpe != NULL, the score-side positional table
 *   pos_a[(q*nh + h), p] = scale * sum_j kpe[h*hd + j, p] * qp[q, h*hd + j]  ( = qt . PE[p] )
 * from kpe = w_k . PE^T (f32 [E, P]; depends on the weights only -- the caller caches it, the way
 * the reference caches its pos_embed buffer, projector.py:603-607).  Rows >= nq*nh of qt_hi /
 * qt_lo / pos_a are not written (callers keep them zero), except that fill_row (bf16 [E], may be
 * NULL) is broadcast into rows [fill_row0, fill_row0 + fill_rows) of qt_hi: the local query rows of
 * hicom_fused_stream_fwd's operand. */
int hicom_fold_query_split_fwd(const float* qp, const void* w_k, const float* kpe, int32_t nq, int32_t nh,
                               int32_t E, int32_t P, float scale, void* qt_hi, void* qt_lo,
                               float* pos_a, int32_t pos_stride, const void* fill_row, int32_t fill_row0,
                               int32_t fill_rows, void* stream);

/* Release recipe, use_guide = "direct" (ONE injected query row = the guide): q_proj, the fold, the score-side positional
 * table, the local query rows of hicom_fused_stream_fwd's operand and the guide-dependent part of the global readout in ONE
 * launch (the q_proj -> fold hand-off travels inside the grid as tagged 8-byte granules):
 *   qt_hi / qt_lo rows h < nh : bf16 hi / lo of  scale * w_k[h]^T (w_q guide + b_q)[h]       (projector.py:180-181,193-197)
 *   qt_hi rows nh .. 15       : local_q bf16 [E], the shared local query (= the guide, :352-368); qt_lo rows stay as
 *                               the caller zeroed them
 *   pos_a[h, p]               : scale * kpe[h]^T (w_q guide + b_q)[h]   (kpe f32 [E, P] = w_k . PE^T, or NULL)
 *   r0 f32 [hidden]           : g_w0 (b_o + guide) + g_b0  -- with C = g_w0 . w_o (cached by the caller) the tail
 *                               GELU(g_w0 (w_o o + b_o + guide) + g_b0) of :226,:646,:307-312 is GELU(C o + r0); g_w0 NULL = no r0
 * state: hicom_query_prep_state_bytes(E) bytes of caller-owned device memory, zeroed ONCE, private to one stream of
 * calls with ONE problem shape (words 0-1 arrival counter, word 2 failed hand-offs, word 3 grid size of the first launch,
 * then the granules; the kernel maintains it).  A hand-off that fails -- a bounded spin that gave up, or a launch whose grid
 * differs from the first one on this block -- writes NaN into qt_hi / qt_lo / pos_a (every output token then is NaN) and
 * counts in word 2: a missed hand-off never passes for a result. */
int64_t hicom_query_prep_state_bytes(int32_t E);
int hicom_query_prep_fwd(const void* guide, const void* local_q, const void* w_q, const void* b_q, const void* w_k, const float* kpe,
                         int32_t nh, int32_t E, int32_t P, float scale, void* qt_hi, void* qt_lo, float* pos_a,
                         int32_t pos_stride, int32_t rows, const void* g_w0, const void* g_b0, const void* b_o,
                         int32_t hidden, float* r0, void* state, void* stream);

/* Split f32 rows into bf16 hi + lo parts (x ~= hi + lo to 2^-16), zero-padding the row count
 * to rows_pad: the MFMA operand format for fp32 intermediates (SURVEY.md §7 strategy B). */
int hicom_split_bf16_fwd(const float* x, int32_t rows, int32_t rows_pad, int32_t E,
                         void* hi, void* lo, void* stream);

/* ---- global compressor: streaming multi-query attention over all tokens -------------------
 * One pass over x = frames_feature[n0 : n0+N) (bf16 [N,E], E = 1152 | 768) computing, for a
 * block of `rows_pad` folded queries (multiple of 16), the raw scores and the online-softmax
 * partial sums of attn . x.  Replaces projector.py:193-215 for K = V = x (pos-emb handled
 * separably: `pos_a` holds qt . PE for the t / y / x axes).
 *   qt_hi, qt_lo : bf16 [rows_pad, E]; rows >= `rows` are zero padding and produce no output
 *   pos_a        : f32 [rows_pad, pos_stride] or NULL; token (t,y,x) adds
 *                  pos_a[r, t_index0 + t] + pos_a[r, y_index0 + y] + pos_a[r, x_index0 + x]
 *   geometry     : token n -> t = n / (H*W), y = (n / W) % H, x = n % W  (n relative to n0 = 0
 *                  of THIS call's x pointer; t_index0 carries the shard's frame offset)
 *   scores       : f32 [rows_pad, score_stride]  (score_stride >= roundup(N,16))
 *   part_m/l     : f32 [nparts, rows_pad]; part_acc : f32 [nparts, rows_pad, E]
 *   nparts       : number of token chunks (= workgroups per row group); returned layout is
 *                  consumed by hicom_global_merge_fwd.
 */
int hicom_global_stream_fwd(const void* x, int64_t N, int32_t E,
                            const void* qt_hi, const void* qt_lo, int32_t rows, int32_t rows_pad,
                            const float* pos_a, int32_t pos_stride,
                            int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                            float* scores, int64_t score_stride,
                            float* part_m, float* part_l, float* part_acc, int32_t nparts,
                            void* stream);
/* Same pass for the clip-scale variant (projector.py:184-191): logit[r, n] = (qt_r . x_n + pos terms + row_const[r]) *
 * inv_norm[n], inv_norm f32 [N] = 1 / ||k_proj(x_n + pos_n)|| (hicom_inv_norm_fwd), row_const f32 [rows_pad]
 * (hicom_clip_query_prep_fwd; padding rows 0).  One 16-row group per workgroup (any row count). */
int hicom_global_stream_clip_fwd(const void* x, int64_t N, int32_t E,
                                 const void* qt_hi, const void* qt_lo, int32_t rows, int32_t rows_pad,
                                 const float* pos_a, int32_t pos_stride,
                                 int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                 const float* inv_norm, const float* row_const,
                                 float* scores, int64_t score_stride,
                                 float* part_m, float* part_l, float* part_acc, int32_t nparts, void* stream);

/* The same pass with the positional marginals of the softmax weights accumulated IN the kernel (projector.py:176-179: the
 * value-side pos-emb term sum_n p_n pos(n) collapses to the t / y / x marginals of p): `scores` may be NULL -- the
 * [rows_pad, N] logit tensor (54 MB at 32 queries x 9 heads x 64 frames) is then neither written nor read back, and
 * hicom_global_merge_marg_fwd needs no per-frame marginal pass.  Only the many-row form of the kernel has it (E = 1152, rows_pad
 * a multiple of 32, grids up to 64 x 48): hicom_global_stream_has_marg(N, E, rows_pad, H, W, nparts) == 1, else HICOM_EUNSUP.
 *   part_marg : f32 [nparts][rows_pad][hicom_global_stream_marg_width(H, W)], un-normalised, relative to the chunk's own max
 *               like part_acc: [8 frames counted from the chunk's first | pad to 16 | H | pad to 16 | W | pad to 16]
 *   pos_a     : required (without pos-emb there are no marginals: use hicom_global_stream_fwd) */
int hicom_global_stream_marg_fwd(const void* x, int64_t N, int32_t E,
                                 const void* qt_hi, const void* qt_lo, int32_t rows, int32_t rows_pad,
                                 const float* pos_a, int32_t pos_stride,
                                 int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                 float* scores, int64_t score_stride,
                                 float* part_m, float* part_l, float* part_acc, float* part_marg, int32_t nparts,
                                 void* stream);
int hicom_global_stream_marg_width(int32_t H, int32_t W);
int hicom_global_stream_has_marg(int64_t N, int32_t E, int32_t rows_pad, int32_t H, int32_t W, int32_t nparts);

/* ---- backward of the global attention over the token stream (training) ------
 * Autograd through projector.py:197-215 in the folded form: with the forward's logits S (the `scores`
 * hicom_global_stream_fwd wrote), its softmax state ml [rows][2] = (M, L) and delta_r = dctx_r . ctx_r,
 *   dP[r,n] = dctx_r . (x_n + pos_n),   dS[r,n] = exp(S[r,n] - M_r) / L_r * (dP[r,n] - delta_r)
 *   part_acc[p, r, :] = sum over chunk p of dS[r,n] x_n          (sum over p = the x part of d qt_r)
 * dctx_hi / dctx_lo: bf16 [rows_pad, E] planes of the upstream context gradients (padding rows zero);
 * pos_b: f32 [rows_pad, pos_stride] = dctx . PE^T (or NULL); ds_out: f32 [rows_pad, score_stride] (its t / y / x
 * marginals times PE give the positional part of d qt).  One pass over x, same tiling as the forward.  part_marg (may be NULL; f32
 * [nparts][rows_pad][hicom_global_stream_marg_width(H, W)], shapes with hicom_global_stream_has_marg == 1): the t / y / x marginals of dS
 * per token chunk, laid out as in hicom_global_stream_marg_fwd (no rescaling: plain sums) -- ds_out may then be NULL and the
 * [rows, N] dS tensor is never written. */
int hicom_global_stream_bwd(const void* x, int64_t N, int32_t E, const void* dctx_hi, const void* dctx_lo,
                            int32_t rows, int32_t rows_pad, const float* pos_b, int32_t pos_stride,
                            int32_t H, int32_t W, int32_t t_index0, int32_t y_index0, int32_t x_index0,
                            const float* s_in, int64_t score_stride, const float* ml, const float* delta,
                            float* ds_out, float* part_acc, float* part_marg, int32_t nparts, void* stream);

/* Suggested nparts for N tokens (fills the chip: 2 workgroups per CU). */
int hicom_global_stream_nparts(int64_t N, int32_t rows_pad);

/* ---- fused local + global stream (release recipe) -------------------------------------------
 * ONE pass over frames_embed and frames_feature computing BOTH the local window contexts
 * (projector.py:544-558, guide as the shared query) and the global online-softmax partial state
 * (projector.py:193-215): each visual tensor is read from HBM exactly once.
 * Requirements: windows partition the grid exactly (T % kt == H % ks == W % ks == 0), 16 <= kt*ks*ks
 * <= 64, rows <= 12 global folded rows, E == 1152.
 *   fe          : frames_embed bf16 [T,H,W,E] -- enters only through the local logit fe_n . local query per token -- or
 *   local_logits: f32 [T*H*W] = fe_n . local query already computed by the producer of frames_embed
 *                 (hicom_dense16_gemm_fwd's row-dot epilogue on the SigLIP head projection, encoder.py:284-286): the
 *                 kernel then does not read frames_embed at all (half the HBM bytes); NULL = stream fe
 *   q_hi / q_lo : bf16 [16, E]; rows < `rows` = folded global queries (hi / lo), rows >= `rows` =
 *                 the local query in q_hi (exact bf16; row `rows` is the one the kernel reads) and zeros in q_lo
 *   pos_a       : f32 [16, pos_stride] score-side pos-emb  a[r, p] = qt_r . pe[p]  (hicom_fold_query_split_fwd),
 *                 or NULL (no pos-emb)
 *   pe_hi/pe_lo : bf16 [P, E] hi / lo planes of the per-axis sinusoid table pe (hicom_split_bf16_fwd of the f32
 *                 table, projector.py:57-101); NULL iff pos_a is NULL.  The value-side pos-emb term
 *                 sum_n p_n (pe[t_index0 + t_n] + pe[y_index0 + y_n] + pe[x_index0 + x_n])  (:176-179) is folded into
 *                 part_acc by the kernel itself (marginals of the weights x pe rows, as extra P.x steps), so the
 *                 partials merge with hicom_global_merge_fwd(pe = NULL).
 *   part_*      : as hicom_global_stream_fwd with rows_pad = 16; nparts from
 *                 hicom_fused_stream_nparts(number of windows)
 *   ctx_local   : f32 [Nw, E], window order (t1,h1,w1), and/or ctx_hi + ctx_lo: the same contexts as
 *                 bf16 planes (hi + lo) for hicom_planes_gemm_fwd, and/or ctx_f16: one fp16 plane (saturating) for
 *                 hicom_readout16_gemm_fwd; unused outputs NULL
 *   part_ctx_f16: not NULL: instead of part_acc (which may then be NULL) the kernel writes the NORMALISED partial contexts
 *                 acc / l as one fp16 plane [nparts][16][E] (rows < rows), for hicom_merge_vproj_fixed_fwd(part_dt = HICOM_DT_F16):
 *                 half the bytes of the partial states; the rounding (2^-12 relative per partial) averages over the partials
 *   zero_ptr    : zero_bytes (multiple of 8, 8-byte aligned) of scratch that the kernel clears for the launches behind it on the
 *                 stream (the fixed-point accumulators of hicom_merge_vproj_fixed_fwd), or NULL
 *   part_marg_f16 (ABI 15): not NULL (with pe_hi = pe_lo = NULL, pos_a and part_ctx_f16 given): the value-side pos-emb term is NOT folded
 *                 into the partial contexts; the kernel writes the t / y / x MARGINALS of every partial's softmax weights instead,
 *                 normalised like part_ctx_f16 (marginal / l), fp16 [nparts][rows][marg_slots] in absolute slot order
 *                 [T frames | H grid rows | W grid columns | zeros], marg_slots >= T + H + W and a multiple of 8 -- for the merge role
 *                 of hicom_readout16_gemm_role_fwd (hicom_r16_role.part_marg), which multiplies the merged marginals by v_proj . pe^T */
int hicom_fused_stream_fwd(const void* ff, const void* fe, const float* local_logits, int32_t T, int32_t H, int32_t W, int32_t E,
                           int32_t kt, int32_t ks, const void* q_hi, const void* q_lo, int32_t rows,
                           float l_scale, float l_bias, const float* pos_a, int32_t pos_stride,
                           const void* pe_hi, const void* pe_lo,
                           int32_t t_index0, int32_t y_index0, int32_t x_index0,
                           float* part_m, float* part_l, float* part_acc, int32_t nparts, float* ctx_local,
                           void* ctx_hi, void* ctx_lo, void* ctx_f16, void* zero_ptr, int64_t zero_bytes, void* part_ctx_f16,
                           void* part_marg_f16, int32_t marg_slots, void* stream);
int hicom_fused_stream_nparts(int32_t n_windows);

/* ---- merge the partials (+ the value-side positional term) --------------------------------
 * out_acc[r,:] = sum_p e^(m_p - M_r) acc_p[r,:] + sum_n e^(s_n - M_r) pos(n)   (un-normalised)
 * out_ml[r]   = (M_r, L_r).  pos(n) = pe[t_index0+t] + pe[y_index0+y] + pe[x_index0+x] with
 * pe f32 [*, E] the per-axis sinusoid tables (projector.py:57-101), or NULL for no pos-emb.
 * scratch: f32 [rows * T * (H+W+2)] work area.  T = N / (H*W).  normalize != 0 divides by L_r
 * (single-shard case: out_acc is then the final context and no combine is needed). */
int hicom_global_merge_fwd(const float* part_m, const float* part_l, const float* part_acc,
                           int32_t nparts, int32_t rows, int32_t rows_pad, int32_t E,
                           const float* scores, int64_t score_stride, int64_t N,
                           int32_t H, int32_t W, const float* pe,
                           int32_t t_index0, int32_t y_index0, int32_t x_index0,
                           float* scratch, float* out_ml, float* out_acc, int32_t normalize, void* stream);

/* Merge behind hicom_global_stream_marg_fwd (same outputs; pe required; scratch as above, only the head of each row's region is
 * used for the row's weight record). */
int hicom_global_merge_marg_fwd(const float* part_m, const float* part_l, const float* part_acc, const float* part_marg,
                                int32_t nparts, int32_t rows, int32_t rows_pad, int32_t E, int64_t N,
                                int32_t H, int32_t W, const float* pe,
                                int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                float* scratch, float* out_ml, float* out_acc, int32_t normalize, void* stream);

/* Combine `nsets` (M,L,ACC) triples (one per GPU after the all-gather, or one) and normalise:
 * ctx[r,:] = sum_k e^(M_k - M) ACC_k[r,:] / sum_k e^(M_k - M) L_k.
 * ml f32 [nsets, rows, 2]; acc f32 [nsets, rows, E]; ctx f32 [rows, E]. */
int hicom_global_combine_fwd(const float* ml, const float* acc, int32_t nsets, int32_t rows,
                             int32_t E, float* ctx, void* stream);

/* ---- readout MLP GEMM on matrix cores -------------------------------------------------------
 * y[out_row(m), :] = act(x[m,:] . w^T + b),  x f32 [M,K] (split hi/lo on the fly), w bf16 [N,K],
 * b bf16|f32 [N].  out_row(m) = row0 + m + (nl_group ? m / nl_group : 0) implements the
 * newline-interleaved packing of post_process_visual_feature (mm_utils.py:100-135).
 * Replaces build_mlp's Linear/GELU/Linear (projector.py:307-312,559,646).
 * K % 64 == 0; y dtype y_dt, row length ldy elements. */
int hicom_readout_gemm_fwd(const float* x, const void* w, const void* b, int32_t b_dt,
                           int32_t M, int32_t N, int32_t K, int32_t act,
                           void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group,
                           void* stream);

/* Same GEMM on bf16 PLANES (the hot path): the activation arrives already split as a_hi + a_lo
 * (bf16 [M,K] each, written by the producing kernel), so all three operands go HBM -> LDS by LDS-DMA.
 * a_lo may be NULL when the activation is exactly bf16 (raw visual tokens).
 * Outputs (either or both): out_hi/out_lo = bf16 planes [M,N] of the result (feeds the next GEMM);
 * y = packed rows as in hicom_readout_gemm_fwd. */
int hicom_planes_gemm_fwd(const void* a_hi, const void* a_lo, const void* w, const void* b, int32_t b_dt,
                          int32_t M, int32_t N, int32_t K, int32_t act, void* out_hi, void* out_lo,
                          void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group, void* stream);

/* ---- hot-path readout GEMM: ONE fp16 plane, with an optional co-scheduled GEMV ---------------------------------
 * y[out_row(m), :] / out_f16[m, :] = act(a[m,:] . w^T + b): a fp16 [M,K] (window contexts / hidden activations written
 * by the producing kernel), w fp16 [N,K] (hicom_to_f16_fwd of the bf16 nn.Linear weight, cached per weight version by the
 * caller), b bf16|f32 [N].  fp16 keeps 11 significand bits of the activation (<= 2^-12 relative, against the 1e-3
 * budget) at half the operand bytes and MFMAs of the hi/lo bf16 planes.  K % 64 == 0.
 * aux (may be NULL): a single-row linear layer executed by extra workgroups of the SAME launch on the CUs the tile grid
 * leaves idle -- the global compressor's out_proj / readout layers ride under the two local readout GEMMs:
 *   y[n] = act(sum_k w[n,k] x[k] + b[n]) + res[n],  x[k] = sum_{s < x_parts} xs[s * x_stride + k] + xb[k]
 * (xb, res bf16; w, b bf16 or f32 (w_dt / b_dt); xs, y f32; K <= 1536, K % 8 == 0). */
typedef struct hicom_aux_gemv {
    const float* xs;
    int32_t x_parts;
    int64_t x_stride;
    const void* xb;
    const void* w;
    const void* b;
    const void* res;
    int32_t N, K, act;
    float* y;                  /* f32 [N]; may be NULL when rows_dst is given */
    int32_t w_dt, b_dt;        /* HICOM_DT_BF16 | HICOM_DT_F32 (f32: products of weight matrices cached by the caller) */
    /* optional: the result row replicated into rows rows_row0 .. rows_row0 + rows_reps - 1 of rows_dst [*, rows_ld]
     * (dtype rows_dt): the 32 identical global rows of "direct" mode (projector.py:646,707) */
    void* rows_dst;
    int32_t rows_dt, rows_reps;
    int64_t rows_ld, rows_row0;
    /* alternative source of x: x[k] = x_fixed[k] / HICOM_FIXED_SCALE + xb[k] (hicom_merge_vproj_fixed_fwd's accumulators); then xs
     * may be NULL and x_parts is ignored */
    const int64_t* x_fixed;
    /* GEMV_CHAIN only (ABI 15): once every role workgroup has read x (= when a workgroup has swept every granule of the first layer),
     * role workgroup 0 clears the x_fixed accumulators for the next step -- the FINISH phase of the frame-sharded step then needs no
     * memset launch in front of its merge.  The accumulators must be zero before the FIRST step. */
    int32_t x_fixed_clear;
} hicom_aux_gemv;
int hicom_readout16_gemm_fwd(const void* a, const void* w, const void* b, int32_t b_dt,
                             int32_t M, int32_t N, int32_t K, int32_t act, void* out_f16,
                             void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group,
                             const hicom_aux_gemv* aux, void* stream);
/* The same launch with a ROLE for the workgroups behind the tile grid (round 5: four launches per step of the release recipe):
 *   HICOM_ROLE_GEMV         `gemv`, as hicom_readout16_gemm_fwd's aux;
 *   HICOM_ROLE_MERGE_VPROJ  the (head, 64-channel slab) items of hicom_merge_vproj_fixed_fwd (fp16 partial contexts only): the merge
 *                           of the ring kernel's partial states (projector.py:193-215 after folding) is independent of the local
 *                           readout (projector.py:559), so it rides under readout GEMM 1 and its launch disappears;
 *   HICOM_ROLE_GEMV_CHAIN   two dependent single-row layers in one launch: h = act(gemv.w . x + gemv.b) with x from gemv.x_fixed,
 *                           then gemv2 with x = h (gemv2.K == gemv.N <= 1536, gemv2.w bf16): GELU(C o + r0) and the last global
 *                           readout layer (projector.py:226, :646, :307-312) under readout GEMM 2.  h travels between the role's
 *                           workgroups as {epoch, value} granules in `chain_state` (hicom_r16_chain_state_bytes(gemv.N) bytes,
 *                           zeroed ONCE by the caller, owned by one plan: every launch on it must have the same tile grid); a
 *                           failed hand-off (bounded spin) writes NaN rows and counts in word 2 of the state block. */
#define HICOM_ROLE_NONE 0
#define HICOM_ROLE_GEMV 1
#define HICOM_ROLE_MERGE_VPROJ 2
#define HICOM_ROLE_GEMV_CHAIN 3
typedef struct hicom_r16_role {
    int32_t kind;
    hicom_aux_gemv gemv;
    hicom_aux_gemv gemv2;
    void* chain_state;
    /* MERGE_VPROJ: the arguments of hicom_merge_vproj_fixed_fwd */
    const float* part_m;
    const float* part_l;
    const void* part_acc;
    int32_t part_dt, nparts, rows, rows_pad, E;
    const void* w_v;           /* NULL (with o_fix NULL): merge only, no v_proj -- the shard state of the frame-sharded path */
    int64_t* o_fix;
    float* out_ml;
    float* out_ctx;
    int32_t ctx_unnorm;        /* out_ctx receives the un-normalised accumulator sum_i e^(m_i - M) ACC_i (with out_ml = (M, L): a shard STATE) */
    /* value-side pos-emb in the merge (ABI 15; replaces `ff += pos` of projector.py:636-640 on the VALUE side): part_marg = fp16
     * [nparts][rows][marg_slots], the NORMALISED t / y / x marginals of every partial's softmax weights in absolute slot order
     * [T | H | W | zero padding] (hicom_fused_stream_fwd's part_marg_f16); vpe_f16 = fp16 [E][marg_slots] = v_proj.weight . pe^T over the
     * same slots (weight-only).  o_fix then receives W_v (ctx + sum_s mg[s] pe[s]).  marg_slots = 8 * (E / 64).  NULL: the partial
     * contexts already carry the pos-emb (or there is none). */
    const void* part_marg;
    const void* vpe_f16;
    int32_t marg_slots;
} hicom_r16_role;
int hicom_readout16_gemm_role_fwd(const void* a, const void* w, const void* b, int32_t b_dt,
                                  int32_t M, int32_t N, int32_t K, int32_t act, void* out_f16,
                                  void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group,
                                  const hicom_r16_role* role, void* stream);
int64_t hicom_r16_chain_state_bytes(int32_t n_mid);
/* The GEMV chain as a launch of its own (no tile grid): role.kind must be HICOM_ROLE_GEMV_CHAIN.  The FINISH phase of the frame-sharded
 * step: GELU(C o + r0) and the last global readout layer behind the merge of the gathered shard states. */
int hicom_gemv_chain_fwd(const hicom_r16_role* role, void* stream);
/* The readout tail as ONE launch (replaces nn.Linear / GELU / nn.Linear of the local readout, projector.py:307-312,559, with the merge of
 * the global partial states, v_proj, out_proj and the global readout, projector.py:226,640-646, riding beside them): GEMM 1's tiles
 * publish the fp16 hidden plane inside the launch, the role workgroups run `merge` then `chain`, GEMM 2's tiles consume the plane row
 * block by row block.  g2->a must be g1->out_f16.  `state`: hicom_readout_tail_state_bytes() bytes, 128-byte aligned, zeroed ONCE
 * (the counters are cumulative); one state block per stream of launches.  HICOM_EUNSUP for shapes outside the fused form (the
 * caller then issues the two hicom_readout16_gemm_role_fwd launches). */
typedef struct hicom_r16_gemm {
    const void* a;             /* fp16 [M, K] */
    const void* w;             /* fp16 [N, K] */
    const void* b;             /* bias [N] (b_dt) or NULL */
    int32_t b_dt, M, N, K, act;
    void* out_f16;             /* fp16 [M, N] plane, or NULL */
    void* y;                   /* packed rows (y_dt), or NULL */
    int32_t y_dt;
    int64_t ldy, row0;
    int32_t nl_group;
} hicom_r16_gemm;
int64_t hicom_readout_tail_state_bytes(void);
int hicom_readout_tail_fwd(const hicom_r16_gemm* g1, const hicom_r16_gemm* g2, const hicom_r16_role* merge, const hicom_r16_role* chain,
                           void* state, void* stream);
/* hicom_merge_vproj_fixed_fwd over shard STATES as hicom_compressor_fwd's STREAM phase leaves them: `nsets` sets, `set_stride` floats apart,
 * each [(M, L) x rows | ACC rows x E] with un-normalised f32 accumulators.  o_fix as above (zero on entry). */
int hicom_merge_vproj_sets_fwd(const float* sets, int64_t set_stride, int32_t nsets, int32_t rows, int32_t E, const void* w_v, int64_t* o_fix,
                               float* out_ml, float* out_ctx, void* stream);
/* dst fp16 [rows, ld_dst] = saturating cast of src (bf16 or f32) [rows, cols]; columns [cols, ld_dst) are zero-filled */
int hicom_to_f16_fwd(const void* src, int32_t src_dt, void* dst, int64_t n, void* stream);
/* ABI 15: contiguous 16-bit <-> 16-bit cast, fp16 -> bf16 (round to nearest even) or bf16 -> fp16 (saturating); 16-byte aligned tensors.
 * The boundary's fp16 width: the reference's inference default is fp16 (inference_video_mcqa_videomme.py:323, model/__init__.py:44,
 * projector.py:53 `load_mm_projector` casts to fp16) -- an fp16 projector runs on the bf16 kernels through casts of its inputs and of its
 * result (hicom_amd/projector.py: HIComProjector._forward_half). */
int hicom_cast16_fwd(const void* src, int32_t src_dt, void* dst, int32_t dst_dt, int64_t n, void* stream);
int hicom_to_f16_padded_fwd(const void* src, int32_t src_dt, int64_t rows, int64_t cols, void* dst, int64_t ld_dst, void* stream);

/* Release recipe, one query row per head: merge of the ring kernel's partial states fused with v_proj
 * (projector.py:182,215 after folding).  po f32 [E/64][E]: partial v_proj outputs per 64-channel slab of the context,
 * summed in slab order by the consumer (hicom_readout16_gemm_fwd's aux GEMV with x_parts = E/64).  out_ml [rows][2] and
 * out_ctx [rows][E] (normalised contexts) may be NULL. */
int hicom_merge_vproj_fwd(const float* part_m, const float* part_l, const float* part_acc, int32_t nparts,
                          int32_t rows, int32_t rows_pad, int32_t E, const void* w_v, float* po,
                          float* out_ml, float* out_ctx, void* stream);
/* The same with the slab sums taken inside the launch: o_fix int64 [E] receives sum_slabs po[slab][:] as FIXED-POINT values
 * (value * 2^36 = HICOM_FIXED_SCALE; integer atomic adds, i.e. bit-identical results whatever the arrival order).  o_fix must be
 * ZERO on entry (hicom_fused_stream_fwd's zero_ptr: the launch in front of this one clears it).  part_dt HICOM_DT_F32: part_acc
 * are the un-normalised fp32 accumulators; HICOM_DT_F16: the normalised fp16 contexts (part_ctx_f16).  The consumer reads o_fix through
 * hicom_aux_gemv.x_fixed: one 9-KB vector instead of E/64 partial vectors per workgroup. */
#define HICOM_FIXED_SCALE 68719476736.0f
int hicom_merge_vproj_fixed_fwd(const float* part_m, const float* part_l, const void* part_acc, int32_t part_dt, int32_t nparts,
                                int32_t rows, int32_t rows_pad, int32_t E, const void* w_v, int64_t* o_fix,
                                float* out_ml, float* out_ctx, void* stream);

/* ---- dense 16-bit MFMA GEMM over all tokens (M = T*729) ------------------------------------------------------------
 * C = epilogue(A[M,K] . W[N,K]^T + b): A, W both fp16 or both bf16 (operand_dt), leading dimensions lda / ldw elements,
 * K % 64 == 0 (pad with zero columns), fp32 accumulation; act NONE | GELU (erf) | GELU_TANH.  Outputs, any subset:
 *   out_f16 [M, ldo] fp16 (saturating): the activated value; columns [N, n_store) are written as zeros (the K padding
 *           of a following GEMM);
 *   pre_f16 [M, ldpre] fp16 (saturating; may be NULL): acc + b BEFORE the activation -- what the activation's backward needs, kept by
 *           the training forward of the adaptor MLPs (needs N % 8 == 0, ldpre % 8 == 0, 16-byte aligned rows);
 *   y [M, ldy] bf16 | f32: value + res[m, n] (res bf16 [M, ldr] or NULL);
 *   ssq f32 [ceil(N/64)][M]: partial row sums of squares of (acc + b) per 64-column slice, summed by the consumer (key norms of the
 *           clip-scale global stage, projector.py:184-186).
 *   row_dot f32 [ceil(N/64)][M]: partial dot products sum_n dot_vec[n] * (value + res)[m, n] per 64-column slice (dot_vec bf16 | f32
 *           [N], 16-byte aligned; needs N % 8 == 0 and 16-byte aligned rows).  With the guide embedding as dot_vec on the fc2
 *           launch of the SigLIP head projection (encoder.py:284-286) the sum over slices IS the local logit frames_embed_n . guide
 *           of projector.py:551 before scaling: y may then be NULL and frames_embed is never written.
 * row_tab (may be NULL): f32 [*, row_tab_ld] table whose rows tab_t0 + m / (H*W), tab_y0 + (m / W) % H, tab_x0 + m % W are
 *           added to row m before bias / ssq / activation: the projected positional embedding W . pos(m) of token m,
 *           separable per axis (projector.py:57-101 through k_proj), so that x + pos is never formed.
 * The matrix-core-bound neighbours of the compressor: the SigLIP pooling-head projection that produces frames_embed
 * (encoder.py:284-286), the k / v adaptor MLPs (projector.py:533-534). */
int hicom_dense16_gemm_fwd(const void* a, int64_t lda, const void* w, int64_t ldw, int32_t operand_dt,
                           const void* b, int32_t b_dt, int32_t M, int32_t N, int32_t K, int32_t act,
                           void* out_f16, int64_t ldo, int32_t n_store, void* pre_f16, int64_t ldpre,
                           void* y, int32_t y_dt, int64_t ldy, const void* res, int64_t ldr,
                           float* ssq, const float* row_tab, int64_t row_tab_ld, int32_t tab_H, int32_t tab_W,
                           int32_t tab_t0, int32_t tab_y0, int32_t tab_x0,
                           const void* dot_vec, int32_t dot_vec_dt, float* row_dot, void* stream);
/* The same for TWO problems of one shape in ONE launch (fp16 outputs only): the same layer of the k and of the v adaptor MLP
 * (projector.py:533-534).  The tiles of both problems share the last, partly filled round of workgroup slots. */
int hicom_dense16_gemm_pair_fwd(const void* a_k, const void* w_k, const void* b_k, void* out_k, void* pre_k,
                                const void* a_v, const void* w_v, const void* b_v, void* out_v, void* pre_v,
                                int64_t lda, int64_t ldw, int32_t operand_dt, int32_t b_dt, int32_t M, int32_t N, int32_t K, int32_t act,
                                int64_t ldo, int32_t n_store, int64_t ldpre, void* stream);
/* TN form: c_parts[s][m][n] = sum over the s-th slice of the Kt rows of a[k][m] * b[k][n] -- the weight gradients dW = dY^T X of the
 * token-stream layers (a = dY [Kt, lda], b = X [Kt, ldb], both token-major, both fp16 or both bf16; fp32 accumulation).  The
 * contraction axis is split into `splits` slices (hicom_dense16_tn_splits proposes a count), each writing its own f32 partial
 * [M][ldc]; hicom_partials_sum_fwd(c_parts, splits, M * ldc, out) adds them in slice order (deterministic).  M, N, lda, ldb
 * multiples of 8, ldc of 4; Kt arbitrary. */
int hicom_dense16_tn_splits(int32_t M, int32_t N, int64_t Kt);
int hicom_dense16_tn_fwd(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t operand_dt, int64_t Kt,
                         int32_t M, int32_t N, float* c_parts, int64_t ldc, int32_t splits, void* stream);
/* out[m] = sum_{s < nparts} parts[s * M + m], in slice order (the row_dot partials of hicom_dense16_gemm_fwd -> the per-token
 * local logits hicom_fused_stream_fwd / hicom_compressor_args.local_logits take). */
int hicom_partials_sum_fwd(const float* parts, int32_t nparts, int64_t M, float* out, void* stream);

/* ---- clip-scale on the global stage (projector.py:184-191) ---------------------------------------------------------
 * hicom_clip_query_prep_fwd: qp f32 [nq, E] (q_proj output) is L2-normalised in place over E; c[q*nh + h] =
 *   scale * sum_j qhat[q, h*hd + j] * b_k[h*hd + j]  (the key-bias part of the logit, which no longer cancels once
 *   every key is divided by its own norm).  b_k bf16 [E] or NULL.
 * hicom_inv_norm_fwd: inv[m] = 1 / sqrt(sum_{s < parts} ssq[s * M + m])  (key norms from hicom_dense16_gemm_fwd's ssq). */
int hicom_clip_query_prep_fwd(float* qp, const void* b_k, int32_t nq, int32_t nh, int32_t E, float scale, float* c, void* stream);
int hicom_inv_norm_fwd(const float* ssq, int32_t parts, int64_t M, float* inv, void* stream);
/* out[m,:] = x[m,:] / ||x[m,:]||_2, bf16 [M, E] -> bf16 [M, E] (E % 8 == 0, E <= 1536): frames_embed of the clip-scale local stage
 * when the k adaptor follows it (projector.py:527-529 in front of :533; without adaptor the window kernel normalises in place). */
int hicom_l2norm_stream_fwd(const void* x, void* out, int64_t M, int32_t E, void* stream);
/* out[m,:] = (1 - alpha) src[m,:] + alpha (LayerNorm_eps(x[m,:]) gamma + beta) over all tokens with 16-byte accesses:
 * x fp16 | bf16 | f32 [M, ldx]; gamma, beta, src (may be NULL) bf16; alpha device scalar or NULL (= 1); out fp16 | bf16
 * [M, E]; E % 8 == 0, E <= 1536.  head.layernorm of encoder.py:284 and the adaptor blend of projector.py:533-534. */
int hicom_ln_stream_fwd(const void* x, int32_t x_dt, int64_t ldx, const void* gamma, const void* beta,
                        const void* src, const void* alpha, int32_t alpha_dt, float eps,
                        void* out, int32_t out_dt, int32_t M, int32_t E, void* stream);

/* Row copy / broadcast with dtype conversion into the packed output:
 *   dst[row0 + i*row_step + (nl_group ? i / nl_group : 0), :] = src[(i % src_rows), :],  i in [0,count)
 * (newline tokens, the 32 identical global rows of "direct" mode, and the standalone
 * post_process_visual_feature packing of mm_utils.py:92-140). */
int hicom_scatter_rows_fwd(const void* src, int32_t src_dt, int32_t src_rows, int32_t ncols,
                           void* dst, int32_t dst_dt, int64_t ldd, int64_t row0, int64_t row_step,
                           int32_t nl_group, int32_t count, void* stream);

/* Rows of `nblocks` equal blocks lying `block_stride_bytes` apart (the per-rank segments of an all-gathered
 * buffer) -> consecutive packed rows of dst starting at row0, skipping one row after every nl_group rows (the
 * newline slots of mm_utils.py:100-135); same dtype on both sides, rows and strides multiples of 16 bytes.
 * Used by the frame-sharded path to place every rank's local tokens with one launch. */
int hicom_place_blocks_fwd(const void* src, int32_t block_rows, int32_t nblocks, int64_t block_stride_bytes,
                           int32_t row_bytes, void* dst, int64_t ldd_bytes, int64_t row0, int32_t nl_group,
                           void* stream);

/* Same, with the per-set (M,L) pairs and ACC blocks `set_stride` floats apart (the layout of the
 * all-gathered per-rank buffers): ml_k = ml + k*set_stride, acc_k = acc + k*set_stride. */
int hicom_global_combine_strided_fwd(const float* ml, const float* acc, int64_t set_stride, int32_t nsets,
                                     int32_t rows, int32_t E, float* ctx, void* stream);

/* ---- instruction injector / adaptor row operators -------------------------------------------
 * out = (1 - alpha) * src + alpha * (LayerNorm_eps(x * (1 + mul) + add) * gamma + beta), row-wise over E.
 * Covers GuideInjector "coarse" (FiLM + LayerNorm, projector.py:369-372: mul/add = one broadcast
 * row, stride 0), "fine" (LayerNorm(q + attn), :392: add per row), and the alpha-blended adaptors
 * (adapt_guide :365, adapt_q :541, adapt_k / adapt_v :533-534).  mul / add / src / alpha may be NULL
 * (alpha == NULL means 1 and needs no src); alpha is a 1-element DEVICE tensor (the nn.Parameter). */
int hicom_row_ln_fwd(const void* x, int32_t x_dt, int64_t x_stride,
                     const float* mul, int64_t mul_stride, const float* add, int64_t add_stride,
                     const void* gamma, const void* beta, int32_t gb_dt,
                     const void* src, int32_t src_dt, int64_t src_stride,
                     const void* alpha, int32_t alpha_dt, float eps,
                     void* out, int32_t out_dt, int64_t out_stride, int32_t M, int32_t E, void* stream);

/* Multi-head attention of M query rows over L <= 64 keys (the text tokens of "fine" injection,
 * projector.py:391 -> :193-215): q [M,E], k, v [L,E] are the PROJECTED states (f32), heads of hd
 * channels, scale hd^-1/2, fp32 softmax; out [M,E] f32 (before out_proj). */
int hicom_small_mha_fwd(const float* q, const float* k, const float* v, int32_t M, int32_t L,
                        int32_t nh, int32_t hd, float* out, void* stream);
/* The same with the logit scale given: the clip-scale form of projector.py:184-191 on the short-key branch passes the
 * L2-normalised projected states (hicom_clip_query_prep_fwd normalises rows in place) and scale = exp(logit_scale); the
 * additive logit_bias is a per-row shift that the softmax cancels. */
int hicom_small_mha_scaled_fwd(const float* q, const float* k, const float* v, int32_t M, int32_t L,
                               int32_t nh, int32_t hd, float scale, float* out, void* stream);

/* ---- splice of the compressed tokens into the LLM input embeddings (hicom_arch.py:271-373) ----------------------------
 * hicom_splice_rows_fwd: dst [nrows, row_bytes] <- row r copied from the DEVICE address row_src[r] (uint64 table on the
 *   device; 0 = a zero row: right padding).  Sources are rows of embed_tokens.weight and of the compressed-token tensors;
 *   the host (hicom_amd/splice.py) plans the table from the token ids.  row_bytes % 16 == 0.
 * hicom_splice_labels_fwd: new_labels int64 [B, Lmax] = labels[b, map[b,p]] where map >= 0, else ignore_index (visual
 *   tokens and padding, :309-311,:344-348); new_mask [B, Lmax] = ones for the first new_len[b] - S positions, then the
 *   original mask row, then zeros (:350-366); mask elements 1 (torch.bool) or 8 (torch.long) bytes.  labels / new_labels or
 *   mask / new_mask may be NULL. */
int hicom_splice_rows_fwd(const void* row_src, int64_t nrows, int32_t row_bytes, void* dst, void* stream);
int hicom_splice_labels_fwd(const void* labels, const void* mask, int32_t mask_elem_bytes, const int32_t* map,
                            const int32_t* new_len, int32_t B, int32_t S, int32_t Lmax, int64_t ignore_index,
                            void* new_labels, void* new_mask, void* stream);

/* ---- whole-forward executor ------------------------------------------------------------------
 * hicom_compressor_fwd enqueues HIComProjector.forward (projector.py:676-708) for one dense
 * [T,H,W,E] input as a fixed plan of kernel launches over a caller-owned workspace.  Release recipe
 * (use_guide = "direct", exact window partition, solo call): everything in order on stream_main -- query prep, the
 * fused stream kernel, merge + v_proj, the two readout GEMMs with the global tail's GEMVs riding in their launches;
 * no side stream, no events.  Generic recipes and the deferred / frame-sharded forms: the local chain on stream_main,
 * the global chain on stream_side (fork/join through ev_fork / ev_join).
 *
 * phases: HICOM_PHASE_STREAM = everything that touches the frames (local tokens, and the
 *         global online-softmax state); HICOM_PHASE_FINISH = the 32 global rows from the state.
 *         A single GPU passes both.  The frame-sharded path runs STREAM with local_out /
 *         state_out pointing into its send buffer, all-gathers, then runs FINISH with state_sets.
 * workspace: hicom_compressor_workspace_bytes(args) bytes, 256-B aligned; its first
 *         hicom_compressor_zero_prefix_bytes(args) bytes must be zeroed ONCE by the caller
 *         (padding rows the kernels never write) and the buffer must not be shared between
 *         concurrently running forwards. */
#define HICOM_PHASE_STREAM 1
#define HICOM_PHASE_FINISH 2
/* With HICOM_PHASE_STREAM on the release recipe, for the frame-sharded step: no side stream at all -- the merge of the
 * partial states runs on stream_next after it has been made to wait for ev_done (both required), i.e. on the comm
 * stream right in front of the all-gather.  Saves the fork / join / ev_merge event traffic (~17 us of host time per
 * step); the caller gives every buffer set its own workspace, since the merge then reads the partials while the
 * main stream may already run the next step. */
#define HICOM_PHASE_MERGE_ON_NEXT 4
/* With HICOM_PHASE_MERGE_ON_NEXT (ABI 15): "next" is this call's own stream_main -- a JOINED frame-sharded step runs both phases on the
 * caller's stream (which may be the null stream, handle 0: stream_next is ignored) and the executor puts no event wait between them. */
#define HICOM_PHASE_NEXT_IS_MAIN 8

typedef struct hicom_compressor_args {
    /* inputs: frames_feature (values; keys/values of the global stage), frames_embed (local keys, may be NULL) */
    const void* ff;
    const void* fe;
    int32_t T, H, W, E;
    int32_t has_local, has_global, phases, hidden;
    /* local compressor */
    hicom_axis at, ay, ax;
    const void* lq;            /* shared [E] (stride 0) or per-window query; NULL = pooled (guide off) */
    int32_t lq_dt;
    int32_t l2norm;
    int64_t lq_stride;
    float l_scale, l_bias;
    const void *lw0, *lb0, *lw2, *lb2;          /* readout Linear(E,hidden), Linear(hidden,hidden): bf16 */
    const void *lw0_f16, *lw2_f16;              /* fp16 copies of lw0 / lw2 (hicom_to_f16_fwd, cached by the caller per weight
                                                   version): the release recipe then runs its readout on one fp16 plane
                                                   (hicom_readout16_gemm_fwd); NULL = bf16 hi/lo planes (hicom_planes_gemm_fwd) */
    /* global compressor */
    const void* gq;            /* bf16 [nq, E] injected queries (1 row in "direct" mode) */
    int32_t nq, nh, n_global_rows, P;
    const void *wq, *bq, *wk, *wv, *bv, *wo, *bo;
    const void *gw0, *gb0, *gw2, *gb2;
    const float* pe;           /* f32 [P, E] stacked per-axis sinusoid tables, or NULL (no pos-emb) */
    const float* kpe;          /* f32 [E, P] = w_k . pe^T (weight-only cache) */
    const void *pe_hi, *pe_lo; /* bf16 [P, E] hi / lo planes of pe (hicom_split_bf16_fwd; cached with pe) */
    int32_t t_index0, y_index0, x_index0, nsets;
    /* output [rows, ldo] of dtype out_dt */
    void* out;
    int32_t out_dt, nl_group;
    int64_t ldo, local_row0, global_row0;
    const void* newline;       /* newline token [hidden] or NULL */
    int32_t newline_dt, nl_count;
    int64_t nl_first, nl_step;
    /* frame-sharded operation (all NULL / 0 on a single GPU) */
    void* local_out;           /* dense [Nw, hidden] (out_dt) instead of packed rows of `out` */
    void* state_out;           /* f32 [2R + R*E]: un-normalised (M,L) pairs then ACC of this shard */
    const void* state_sets;    /* gathered states, nsets blocks state_set_stride floats apart */
    int64_t state_set_stride;
    /* execution resources */
    void* ws;
    int64_t ws_bytes;
    void *stream_main, *stream_side, *ev_fork, *ev_join;
    /* Deferred join (release recipe only): with defer_join != 0 the main stream does NOT wait for the side stream's
     * global chain at the end of the call -- the 32 global rows of `out` are complete when ev_join (recorded on the
     * side stream) has fired, so the chain of one video overlaps the streaming of the next.  ev_merge (may be NULL
     * when defer_join == 0) is recorded after the merge kernel and waited on before the next stream kernel, which
     * overwrites the partial states the merge reads.  Generic path with many query rows (guide off: 288): when given, the side
     * stream launches the stream kernel first and records ev_merge behind it, the main stream waits for it before the local
     * chain -- the stream kernel runs alone, the local chain beside the global tail. */
    void* ev_merge;
    int32_t defer_join, reserved_;
    /* Frame-sharded serving (hicom_amd/dist.py): what follows a phase on its own stream, so that one C call enqueues
     * it all (each separate host call costs 3-6 us, and the sharded step is host-bound).
     *   place_src != NULL (FINISH phase): the gathered per-rank token blocks -- place_nblocks blocks of
     *     place_block_rows rows of `hidden` elements, place_block_stride bytes apart -- are placed into the packed
     *     rows of `out` from row 0 (newline gaps as in nl_group), as hicom_place_blocks_fwd does;
     *   ev_done != NULL: recorded at the end of the call on the stream the phase ran on (after the join);
     *   stream_next != NULL (needs ev_done): that stream is made to wait for ev_done. */
    const void* place_src;
    int64_t place_block_stride;
    int32_t place_block_rows, place_nblocks;
    void *ev_done, *stream_next;
    /* gc0: f32 [hidden, E] = gw0 . wo, the first global readout layer folded over out_proj (weight-only, cached by the
     * caller per weight state like kpe), or NULL.  With it the release recipe's step is five launches:
     * hicom_query_prep_fwd | fused stream | merge + v_proj | readout GEMM 1 (+ GELU(gc0 o + r0)) | readout GEMM 2 (+ the
     * last global readout layer -> the 32 global rows). */
    const float* gc0;
    /* local_logits: f32 [T*H*W] = frames_embed_n . guide per token, or NULL (release recipe only; see hicom_fused_stream_fwd) */
    const float* local_logits;
    /* reuse_queries != 0 (generic recipes, guide off only): the folded queries / score-side positional table in the workspace are
     * those of an earlier call with the SAME weights -- the injected queries are the learnable `query` parameter (IdentityMap,
     * projector.py:586-587), so q_proj + fold are weight-only work like kpe -- and the two prep launches are skipped.  The caller
     * clears it after a weight update. */
    int32_t reuse_queries;
    /* k / v adaptors of the local stage (projector.py:431-457, :533-534: key = (1 - a) x + a LN(MLP(x)) over ALL tokens; likewise
     * the values), the second released recipe `local43_adaptkv_global32`.  ak.w0 / av.w0 NULL = that stream has no adaptor.  The
     * executor runs the two dense GEMMs per stream (hicom_dense16_gemm_fwd) and the window attention with the LayerNorm + alpha
     * blend fused into its row loads (hicom_local_attn_adapt_fwd); not with clip-scale (l2norm != 0) and not on the release
     * recipe's single-kernel path (the local stage then reads adapted streams, the global stage the raw tokens). */
    struct hicom_adaptor {
        const void *w0, *b0;       /* Linear(E, E): bf16 [E, E], bf16 [E] */
        const void *w2_f16, *b2;   /* Linear(E, E): fp16 copy of the weight (cached by the caller per weight state), bf16 bias */
        const void *gamma, *beta;  /* LayerNorm: bf16 [E] */
        const void* alpha;         /* device scalar, dtype adapt_alpha_dt */
        const void* y;             /* optional: fp16 [T*H*W, E] = MLP(x) of this stream, already computed by the caller (the training
                                    * forward keeps the MLP's intermediates for its backward): the adaptor's two GEMMs are skipped */
    } ak, av;
    int32_t adapt_alpha_dt;
    float adapt_eps;
    /* r0_buf (frame-sharded release recipe, may be NULL): f32 [hidden] in CALLER memory that carries r0 = G0 (b_o + g) + g_b0 from the
     * STREAM phase (query prep writes it there) to the FINISH phase (the chain launch reads it), whose workspaces are separate.  With it
     * (and gc0, one query row per head, hidden <= 1536) the sharded step takes the four-launch form: the shard's state (M, L, ACC)
     * comes out of the merge ROLE of readout GEMM 1's launch, and FINISH is [memset | merge of the gathered states + v_proj |
     * chain launch -> 32 rows | token placement]. */
    float* r0_buf;
    /* dtype of gq (HICOM_DT_BF16 = 0, the learnable queries / the guide, or HICOM_DT_F32: queries a guide injector produced on the
     * caller's side -- coarse / fine injection, projector.py:369-397 -- handed in as f32 rows; with lq f32 [windows, E] the same way).
     * The release-recipe paths (query_prep kernel, fused stream kernel) take bf16 queries only. */
    int32_t gq_dt;
    /* optional hipEvent_t: the main stream waits for it in front of the local stage's first launch -- the caller produces lq (injected
     * per-window queries) on a stream of its own, beside the global stage's stream kernel, and records this event behind it */
    void* ev_queries;
    /* GuideInjector.forward inside the call (projector.py:369-397), plain injectors (no text2qk projection, no adapt_guide): the local
     * stage injects the guide into its pooled per-window queries (lq must be NULL), the global stage into `visual` = its learnable
     * queries (gq is ignored).  STREAM | FINISH in one call only (the injected rows live in this call's workspace). */
    struct hicom_injector {
        int32_t mode;              /* 0: none; 1: coarse = LN(v (1 + scale) + shift), (scale | shift) = MLP(guide) (:370-372);
                                    * 2: fine = LN(v + MHA(v, guide tokens)) (:391-392) */
        const void* guide;         /* bf16 [guide_rows, E]: one row (coarse) or <= 64 text tokens (fine) */
        int32_t guide_rows;
        const void *c_w0, *c_b0, *c_w2, *c_b2;       /* coarse_proj: Linear(E, c_hidden) GELU Linear(c_hidden, 2E), bf16 */
        int32_t c_hidden;
        const void *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;   /* fine_proj (MultiheadAttention, :166-228), bf16 [E, E] / [E] */
        int32_t nheads;
        const void *ln_w, *ln_b;   /* coarse_norm / fine_norm, bf16 [E] */
        float eps;
        const void* visual;        /* global stage: bf16 [nq, E]; local stage: NULL */
    } inj_l, inj_g;
    /* ABI 15 (may be NULL / 0): fp16 [E][marg_slots] = v_proj.weight . pe^T over the slots [T frames | H rows | W columns | zero padding],
     * marg_slots = 8 * (E / 64) >= T + H + W: the release step then takes the value-side pos-emb out of the streaming kernel (its
     * marginals leave it, hicom_fused_stream_fwd's part_marg_f16) and applies it in the merge role of readout GEMM 1's launch
     * (hicom_r16_role.part_marg / vpe_f16).  Weight-only, cached by the caller per weight state like kpe. */
    const void* vpe_f16;
    int32_t marg_slots;
    /* ABI 15, FINISH phase of the frame-sharded step (all NULL / 0 otherwise): the all-gather of the per-rank exchange buffers enqueued
     * BY THIS CALL, on stream_main, in front of everything else -- `ag_fn` = address of ncclAllGather of the RCCL the process has loaded
     * (ncclResult_t (*)(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t, hipStream_t)), `ag_comm` = the
     * ncclComm_t of the caller's process group (torch: ProcessGroupNCCL._comm_ptr()), ag_bytes per rank as ncclUint8.  The library does not
     * link RCCL: one host call per phase instead of a torch.distributed call between two (the sharded step is host-bound). */
    void* ag_fn;
    void* ag_comm;
    const void* ag_send;
    void* ag_recv;
    int64_t ag_bytes;
    /* ... or TWO all-gathers in one RCCL group (ag_group_start / ag_group_end = addresses of ncclGroupStart / ncclGroupEnd): the second
     * one (ag_send2 -> ag_recv2, ag_bytes2 per rank) carries the shard STATES into a [world][ag_bytes2] buffer (= state_sets), the first
     * the local token rows STRAIGHT into their rows of the output (ag_recv = out: without newline rows the ranks' blocks are consecutive
     * rows) -- no placement launch behind the collective (place_src NULL). */
    void* ag_group_start;
    void* ag_group_end;
    const void* ag_send2;
    void* ag_recv2;
    int64_t ag_bytes2;
} hicom_compressor_args;

/* Byte offset, inside the workspace, of the fp16 plane [windows, E] of the local stage's window contexts (the A operand of readout
 * GEMM 1) that a hicom_compressor_fwd call with these arguments leaves behind -- what the training forward keeps for the backward
 * (the readout's weight gradients need the contexts; recomputing them is a pass over every token).  Negative: HICOM_EUNSUP when the
 * call does not read out through fp16 planes, HICOM_EINVAL without a local stage. */
int64_t hicom_compressor_ctx16_offset(const hicom_compressor_args* args);
/* Two calls of hicom_compressor_fwd in one (the STREAM and the FINISH block of a frame-sharded step; stops at the first error). */
int hicom_compressor_fwd2(const hicom_compressor_args* first, const hicom_compressor_args* second);
/* 1 when a STREAM-phase call with these arguments takes the four-launch sharded form that hands r0 to the FINISH phase through r0_buf,
 * 0 otherwise (the caller then clears r0_buf in BOTH blocks so that FINISH takes its generic form too: ADVICE r5). */
int hicom_compressor_takes_shard4(const hicom_compressor_args* args);
/* Failed in-launch hand-offs (bounded spins that expired: the affected rows were poisoned with NaN) counted so far in this workspace's
 * state blocks: out[0] = query prep, out[1] = the GEMV chain.  Synchronous (two 4-byte device reads on `stream` + a stream sync):
 * for tests, benches and debug checks, not for the hot loop. */
int hicom_compressor_handoff_failures(const hicom_compressor_args* args, int32_t* out, void* stream);
/* 1 when hicom_compressor_fwd takes the release-recipe (single streaming kernel) path for these arguments. */
int hicom_compressor_is_fused(const hicom_compressor_args* args);

int64_t hicom_compressor_workspace_bytes(const hicom_compressor_args* args);
int64_t hicom_compressor_zero_prefix_bytes(const hicom_compressor_args* args);
int hicom_compressor_fwd(const hicom_compressor_args* args);

#ifdef __cplusplus
}
#endif
#endif /* HICOM_HIP_H */
