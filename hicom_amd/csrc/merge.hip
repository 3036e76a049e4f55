// Merge of the streaming partials of the global compressor, the value-side positional term,
// and the cross-shard (cross-GPU) combine.
//
//   M_r = max_p m_p,  L_r = sum_p e^(m_p - M_r) l_p
//   ACC[r,:] = sum_p e^(m_p - M_r) acc_p[r,:]  +  sum_n e^(s_n - M_r) pos(n)
// with pos(n) = PE_t(t) + PE_y(y) + PE_x(x) (reference projector.py:95-99): the sum over tokens
// collapses to the t / y / x marginals of the softmax weights, so x + pos is never formed.
//
// Two launches: (1) per (row, frame) marginals against the FRAME's own max (independent of the
// partials, so it needs no prior statistics pass), (2) one workgroup per (row, 64-channel slab)
// that derives (M, L) itself, folds the frame marginals and reduces the partial contexts with
// 4-way split over the partial index (coalesced 256-B row segments).
#include <stdlib.h>

#include "common.hpp"
#include "merge_item.hpp"

namespace hicom {

// scratch[(r*T + t) * (H + W + 2) + {0..H-1: y-marginal | H..H+W-1: x-marginal | H+W: total | H+W+1: frame max}]
__global__ __launch_bounds__(256) void frame_marginals_kernel(const float* scores, long score_stride,
                                                              int T, int H, int W, float* scratch) {
    extern __shared__ float e[];   // [H*W] + [H] + [4]
    const int r = blockIdx.x, t = blockIdx.y, HW = H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* s = scores + (long)r * score_stride + (long)t * HW;
    float* fy = e + HW;
    float* wred = fy + H;
    // the frame's logits stay in registers between the max and the exp pass (H * W <= 1024: 4 per thread, loads in flight
    // together); larger grids re-read them
    float sv[4];
    float mx = -1.0e30f;
    const bool small = HW <= 1024;
    if (small) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u;
            sv[u] = i < HW ? s[i] : -1.0e30f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, sv[u]);
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) mx = fmaxf(mx, s[i]);
    }
    mx = wave_max_fast(mx);
    if (lane == 0) wred[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    if (small) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < HW) e[i] = expf(sv[u] - mx);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) e[i] = expf(s[i] - mx);
    }
    __syncthreads();
    float* out = scratch + ((long)r * T + t) * (H + W + 2);
    for (int j = threadIdx.x; j < H + W; j += 256) {
        float a = 0.f;
        if (j < H) {
            for (int x = 0; x < W; ++x) a += e[j * W + x];
            fy[j] = a;
        } else {
            for (int y = 0; y < H; ++y) a += e[y * W + (j - H)];
        }
        out[j] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int y = 0; y < H; ++y) a += fy[y];
        out[H + W] = a;
        out[H + W + 1] = mx;
    }
}

// Wave form for grids up to 32 x 32 (27 x 27 here): one wave per (row, frame), no LDS and no block barrier.  Half-wave h owns
// the grid rows y = 2k + h, lane x of a half the grid column x: all of a lane's <= 16 logits are loaded together, the column
// marginal is a register sum (+ one cross-half shuffle), a row marginal a 32-lane butterfly.  The block form above keeps
// 256 threads around three barriers for 54 short serial sums: 37 us for 288 x 64 frames; it stays for larger grids.
__global__ __launch_bounds__(256) void frame_marginals_wave_kernel(const float* scores, long score_stride,
                                                                   int R, int T, int H, int W, float* scratch) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long)R * T) return;
    const int r = (int)(wid / T), t = (int)(wid - (long)r * T);
    const int half = lane >> 5, x = lane & 31;
    const float* s = scores + (long)r * score_stride + (long)t * H * W;
    float v[16];
    float mx = -1.0e30f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int y = 2 * k + half;
        v[k] = (x < W && y < H) ? s[y * W + x] : -1.0e30f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) mx = fmaxf(mx, v[k]);
    mx = wave_max_fast(mx);
    float* out = scratch + ((long)r * T + t) * (H + W + 2);
    float xm = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int y = 2 * k + half;
        const float e = (x < W && y < H) ? expf(v[k] - mx) : 0.f;
        xm += e;
        const float ys = half32_sum(e);                       // (DPP + row swap: a shuffle butterfly is 5 LDS round trips per row)
        if (x == 0 && y < H) out[y] = ys;
    }
    {
        auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(xm), __float_as_uint(xm), false, false);
        xm = __uint_as_float(b[0]) + __uint_as_float(b[1]);   // both halves of the column
    }
    if (half == 0 && x < W) out[H + x] = xm;
    float tot = half == 0 ? xm : 0.f;
    tot = wave_sum_fast(tot);
    if (lane == 0) {
        out[H + W] = tot;
        out[H + W + 1] = mx;
    }
}

struct MergeCtxParams {
    const float* part_m;
    const float* part_l;
    const float* part_acc;
    int nparts, rows_pad, E;
    const float* scratch;   // frame marginals (two-kernel path) or NULL
    const float* pe;
    int T, H, W, t0i, y0i, x0i;
    float* out_ml;
    float* out_acc;
    int normalize;
};

__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    v = wave_max_fast(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = wave_sum_fast(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int kTC = 64;   // frames of marginals staged in LDS per pass (one pass for T <= 64)

// MODE 0: one kernel does it all (no positional term: the weights are a few exps per workgroup).
// With the positional term the row's weights (M, L, partial weights, the T + H + W positional weights out of the frame
// marginals) cost more than the 32-column slab they are applied to, and every one of the E / 32 workgroups of a row used to
// recompute them: MODE 1 (grid = rows) computes them once and leaves them at the head of the row's scratch region, MODE 2
// (grid = rows x E / 32) applies them.
template <int MODE>
__global__ __launch_bounds__(256) void merge_ctx_kernel(MergeCtxParams p) {
    // LDS: [16*64] column partials | [4] | [nparts] partial weights | [T] frame weights |
    //      [T+H+W] positional weights | [kTC * S] staged marginals
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int S = p.H + p.W + 2, HW2 = p.H + p.W;
    float* cred = wsm;                                   // 16-byte aligned: float4 stores
    float* red = cred + 16 * 64;
    float* wp = red + 4;
    float* wt = wp + p.nparts;
    float* wpos = wt + p.T;
    float* tile = wpos + (p.T + HW2);
    const float* sc = p.scratch ? p.scratch + (long)r * p.T * S : nullptr;

    // Everything that does not depend on M is requested first, so the dependent chain below is
    // M -> weights -> sums instead of one memory round trip per step:
    //   (a) the first batch of partial-context rows of this thread's column group (raw values)
    //   (b) the first chunk of the frame-marginal table -> LDS
    const int pgp = tid >> 3, l4 = tid & 7;
    const int c4 = blockIdx.y * 32 + 4 * l4;
    const float* base = p.part_acc + (long)r * p.E + c4;
    const long pstride = (long)p.rows_pad * p.E;
    float4 v0[8];
    const bool have0 = MODE != 1 && c4 < p.E && pgp + 32 * 7 < p.nparts;
    if (have0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v0[u] = *reinterpret_cast<const float4*>(base + (long)(pgp + 32 * u) * pstride);
    }
    float M, L;
    const bool has_pos = sc != nullptr;
    if constexpr (MODE == 2) {
        // weights left by the MODE 1 launch: [M, L | nparts partial weights | T + H + W positional weights]
        float* wrow = const_cast<float*>(sc);
        for (int i = tid; i < p.nparts; i += 256) wp[i] = wrow[2 + i];
        for (int i = tid; i < p.T + HW2; i += 256) wpos[i] = wrow[2 + p.nparts + i];
        M = wrow[0];
        L = wrow[1];
    } else {
    if (sc) {
        const int nt = min(kTC, p.T);
        for (int i = tid; i < nt * S; i += 256) tile[i] = sc[i];
    }

    // (M, L) of this row from the partials
    float mx = -1.0e30f;
    for (int i = tid; i < p.nparts; i += 256) mx = fmaxf(mx, p.part_m[(long)i * p.rows_pad + r]);
    M = block_reduce_max(mx, red);
    float l = 0.f;
    for (int i = tid; i < p.nparts; i += 256) {
        const float w = expf(p.part_m[(long)i * p.rows_pad + r] - M);
        wp[i] = w;
        l += w * p.part_l[(long)i * p.rows_pad + r];
    }
    L = block_reduce_sum(l, red);

    if (sc) {
        float ay[4] = {0.f, 0.f, 0.f, 0.f};      // up to 4 * 256 spatial marginals per thread
        for (int t0 = 0; t0 < p.T; t0 += kTC) {
            const int nt = min(kTC, p.T - t0);
            if (t0 > 0) {
                __syncthreads();
                for (int i = tid; i < nt * S; i += 256) tile[i] = sc[(long)t0 * S + i];
                __syncthreads();
            }
            if (tid < nt) {
                const float w = expf(tile[tid * S + HW2 + 1] - M);
                wt[t0 + tid] = w;
                wpos[t0 + tid] = w * tile[tid * S + HW2];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = tid + 256 * u;
                if (j < HW2)
                    for (int tt = 0; tt < nt; ++tt) ay[u] = fmaf(wt[t0 + tt], tile[tt * S + j], ay[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = tid + 256 * u;
            if (j < HW2) wpos[p.T + j] = ay[u];
        }
    }
    }   // MODE != 2
    __syncthreads();
    if constexpr (MODE == 1) {
        // every read of the row's scratch region is behind the barrier above: its head becomes the weight record
        float* wrow = const_cast<float*>(sc);
        for (int i = tid; i < p.nparts; i += 256) wrow[2 + i] = wp[i];
        for (int i = tid; i < p.T + HW2; i += 256) wrow[2 + p.nparts + i] = wpos[i];
        if (tid == 0) {
            wrow[0] = M;
            wrow[1] = L;
            p.out_ml[2 * r] = M;
            p.out_ml[2 * r + 1] = L;
        }
        return;
    }

    // context columns: 32 partial-groups x 8 lanes x float4 (32-column slab -> R * E/32 workgroups fill
    // the chip), 8 independent loads in flight per thread
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < p.E) {
        int i = pgp;
        if (have0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float w = wp[i + 32 * u];
                a.x = fmaf(w, v0[u].x, a.x); a.y = fmaf(w, v0[u].y, a.y); a.z = fmaf(w, v0[u].z, a.z); a.w = fmaf(w, v0[u].w, a.w);
            }
            i += 32 * 8;
        }
        for (; i + 32 * 7 < p.nparts; i += 32 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (long)(i + 32 * u) * pstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float w = wp[i + 32 * u];
                a.x = fmaf(w, v[u].x, a.x); a.y = fmaf(w, v[u].y, a.y); a.z = fmaf(w, v[u].z, a.z); a.w = fmaf(w, v[u].w, a.w);
            }
        }
        for (; i < p.nparts; i += 32) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long)i * pstride);
            const float w = wp[i];
            a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
        }
        if (has_pos) {
            // four pe rows of this thread in flight together (T + H + W = 118 at C2: one batch)
            const int npos = p.T + HW2;
            for (int j0 = pgp; j0 < npos; j0 += 128) {
                float4 v[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + 32 * u;
                    const int jc = j < npos ? j : npos - 1;
                    const int row = jc < p.T ? p.t0i + jc : (jc < p.T + p.H ? p.y0i + (jc - p.T) : p.x0i + (jc - p.T - p.H));
                    v[u] = *reinterpret_cast<const float4*>(p.pe + (long)row * p.E + c4);
                    w[u] = j < npos ? wpos[jc] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a.x = fmaf(w[u], v[u].x, a.x); a.y = fmaf(w[u], v[u].y, a.y); a.z = fmaf(w[u], v[u].z, a.z); a.w = fmaf(w[u], v[u].w, a.w);
                }
            }
        }
    }
    *reinterpret_cast<float4*>(cred + pgp * 32 + 4 * l4) = a;
    __syncthreads();
    if (tid < 32) {
        const int c = blockIdx.y * 32 + tid;
        if (c < p.E) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 32; ++g) v += cred[g * 32 + tid];
            if (p.normalize) v /= L;
            p.out_acc[(long)r * p.E + c] = v;
        }
    }
    if (MODE == 0 && blockIdx.y == 0 && tid == 0) {
        p.out_ml[2 * r] = M;
        p.out_ml[2 * r + 1] = L;
    }
}

// MODE 2's job for RG rows at once (round 6): apply the weight records [M, L | nparts partial weights | T + H + W positional weights]
// that MODE 1 / merge_marg_weights_kernel left at the head of each row's scratch region.  MODE 2 ran one workgroup per (row, 32-column
// slab) -- 10 368 at 288 rows -- and each walked a dependent chain of three memory round trips (weight record -> its ONE partial row per
// thread at 32 partials -> four pe rows), re-reading the 15-KB pe slab that every row of the slab shares: 24-33 us for 43 MB
// (profiles/r05_d_recipe_traces.txt).  Here a workgroup owns RG rows x 32 columns: the pe rows of a thread's columns are loaded ONCE into
// registers and serve all RG rows, the RG weight records and the thread's RG partial rows are requested together with them -- one round
// trip.  (The per-output summation order differs from MODE 2's -- 16 groups instead of 32 -- by fp32 re-association only.)
template <int RG>
__global__ __launch_bounds__(256) void merge_ctx_apply_kernel(MergeCtxParams p, int rows) {
    // a workgroup owns RG rows x 64 columns: 16 lanes x float4 across the columns (256 contiguous bytes per partial row: whole DRAM
    // bursts; 32-column slabs read 128-byte pieces 4.6 KB apart), 16 partial groups down the partials / positional rows
    constexpr int CW = 64, NL = CW / 4, NGR = 256 / NL;      // columns, lanes across them, groups
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, r0 = blockIdx.x * RG;
    const int S = p.H + p.W + 2, HW2 = p.H + p.W, npos = p.T + HW2, rec = 2 + p.nparts + npos;
    float* cred = wsm;                                   // [RG][NGR groups][CW columns]
    float* wrec = cred + RG * NGR * CW;                  // [RG][rec]
    const int pgp = tid / NL, l4 = tid % NL;
    const int c4 = blockIdx.y * CW + 4 * l4;
    const bool col_ok = c4 < p.E;
    const long pstride = (long)p.rows_pad * p.E;
    constexpr int PV = 2, PE = 8;                        // partials / positional rows a thread keeps in registers (2 x 16 = 32, 8 x 16 = 128)
    // ---- every request first: the thread's first PV partial rows of each of the RG rows, its PE pe rows, the weight records ----
    float4 v0[RG][PV];
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
        const int r = r0 + rr;
#pragma unroll
        for (int u = 0; u < PV; ++u) {
            const int i = pgp + NGR * u;
            v0[rr][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && col_ok && i < p.nparts) v0[rr][u] = *reinterpret_cast<const float4*>(p.part_acc + (long)i * pstride + (long)r * p.E + c4);
        }
    }
    float4 pev[PE];
#pragma unroll
    for (int u = 0; u < PE; ++u) {
        const int j = pgp + NGR * u, jc = j < npos ? j : npos - 1;
        const int row = jc < p.T ? p.t0i + jc : (jc < p.T + p.H ? p.y0i + (jc - p.T) : p.x0i + (jc - p.T - p.H));
        pev[u] = col_ok ? *reinterpret_cast<const float4*>(p.pe + (long)row * p.E + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int idx = tid; idx < RG * rec; idx += 256) {
        const int rr = idx / rec, k = idx - rr * rec, r = r0 + rr;
        wrec[idx] = r < rows ? p.scratch[(long)r * p.T * S + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
        const int r = r0 + rr;
        const float* w = wrec + rr * rec;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows && col_ok) {
#pragma unroll
            for (int u = 0; u < PV; ++u) {
                const int i = pgp + NGR * u;
                const float wu = i < p.nparts ? w[2 + i] : 0.f;
                a.x = fmaf(wu, v0[rr][u].x, a.x); a.y = fmaf(wu, v0[rr][u].y, a.y); a.z = fmaf(wu, v0[rr][u].z, a.z); a.w = fmaf(wu, v0[rr][u].w, a.w);
            }
            for (int i = pgp + NGR * PV; i < p.nparts; i += NGR) {       // (more than 32 partials: the rest, one at a time)
                const float4 v = *reinterpret_cast<const float4*>(p.part_acc + (long)i * pstride + (long)r * p.E + c4);
                const float wu = w[2 + i];
                a.x = fmaf(wu, v.x, a.x); a.y = fmaf(wu, v.y, a.y); a.z = fmaf(wu, v.z, a.z); a.w = fmaf(wu, v.w, a.w);
            }
            const float* wpos = w + 2 + p.nparts;
#pragma unroll
            for (int u = 0; u < PE; ++u) {
                const int j = pgp + NGR * u;
                const float wu = j < npos ? wpos[j] : 0.f;
                a.x = fmaf(wu, pev[u].x, a.x); a.y = fmaf(wu, pev[u].y, a.y); a.z = fmaf(wu, pev[u].z, a.z); a.w = fmaf(wu, pev[u].w, a.w);
            }
            for (int j = pgp + NGR * PE; j < npos; j += NGR) {            // (more than 128 positional rows: long clips)
                const int row = j < p.T ? p.t0i + j : (j < p.T + p.H ? p.y0i + (j - p.T) : p.x0i + (j - p.T - p.H));
                const float4 v = *reinterpret_cast<const float4*>(p.pe + (long)row * p.E + c4);
                const float wu = wpos[j];
                a.x = fmaf(wu, v.x, a.x); a.y = fmaf(wu, v.y, a.y); a.z = fmaf(wu, v.z, a.z); a.w = fmaf(wu, v.w, a.w);
            }
        }
        *reinterpret_cast<float4*>(cred + (rr * NGR + pgp) * CW + 4 * l4) = a;
    }
    __syncthreads();
    for (int o = tid; o < RG * CW; o += 256) {
        const int rr = o / CW, cc = o % CW, r = r0 + rr, c = blockIdx.y * CW + cc;
        if (r < rows && c < p.E) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < NGR; ++g) v += cred[(rr * NGR + g) * CW + cc];
            if (p.normalize) v /= wrec[rr * rec + 1];
            p.out_acc[(long)r * p.E + c] = v;
        }
    }
}

// launches the apply step: RG = 8 rows per workgroup when the records fit the LDS budget, else MODE 2 (one row per workgroup)
static void launch_merge_apply(const MergeCtxParams& p, int rows, size_t smem2, hipStream_t s) {
    constexpr int RG = 8;
    const size_t rec = (size_t)2 + p.nparts + p.T + p.H + p.W;
    const size_t smem = ((size_t)RG * 16 * 64 + RG * rec) * 4;
    // (the many-row stage: >= 64 rows of <= 32 partials each.  Few rows of many partials -- the direct recipe's 9 rows x 256 partials in the
    // training path -- keep MODE 2: 324 workgroups with eight partial rows in flight per thread; this kernel would run them on 36
    // workgroups, one partial at a time behind the first two: 84 us instead of 9)
    if (smem <= 64 * 1024 && p.scratch && p.pe && rows >= 64 && p.nparts <= 32) {
        HICOM_LAUNCH(merge_ctx_apply_kernel<RG>, dim3((unsigned)((rows + RG - 1) / RG), (unsigned)((p.E + 63) / 64)), dim3(256), smem, s, p, rows);
    } else {
        HICOM_LAUNCH(merge_ctx_kernel<2>, dim3((unsigned)rows, (unsigned)((p.E + 31) / 32)), dim3(256), smem2, s, p);
    }
}

// Row weights from the marginals the wide stream kernel accumulated itself (hicom_global_stream_marg_fwd): grid = rows.  Leaves
// the record merge_ctx_kernel<2> applies -- [M, L | nparts partial weights | T + H + W positional weights] -- at the head of the
// row's scratch region.  part_marg[i][r] holds chunk i's marginals relative to the chunk's own max m_i, like part_acc: the
// weight e^(m_i - M) of the partial contexts applies to them too; its frame block starts at the chunk's first frame.
__global__ __launch_bounds__(256) void merge_marg_weights_kernel(MergeCtxParams p, const float* part_marg, int MS, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int S = p.H + p.W + 2, HW2 = p.H + p.W;
    float* red = wsm;
    float* wp = red + 4;
    int* ff = reinterpret_cast<int*>(wp + p.nparts);       // first frame of each chunk
    float mx = -1.0e30f;
    for (int i = tid; i < p.nparts; i += 256) mx = fmaxf(mx, p.part_m[(long)i * p.rows_pad + r]);
    const float M = block_reduce_max(mx, red);
    float l = 0.f;
    for (int i = tid; i < p.nparts; i += 256) {
        const float w = expf(p.part_m[(long)i * p.rows_pad + r] - M);
        wp[i] = w;
        l += w * p.part_l[(long)i * p.rows_pad + r];
        const int tb = (int)(((long)ntiles * i) / p.nparts);
        ff[i] = (int)((unsigned)(tb * 16) / (unsigned)(p.H * p.W));
    }
    const float L = block_reduce_sum(l, red);
    __syncthreads();
    float* wrow = const_cast<float*>(p.scratch) + (long)r * p.T * S;
    const float* mg = part_marg + (long)r * MS;
    const long mstride = (long)p.rows_pad * MS;
    const int ybase = 16, xbase = 16 + 16 * ((p.H + 15) >> 4);
    // two threads per output (even / odd chunks), loads of 8 chunks in flight together (the frame block only counts where the
    // chunk covers the frame: clamped column, zero weight elsewhere -- no branch around the load)
    for (int j2 = tid; j2 < 2 * (p.T + HW2); j2 += 256) {
        const int j = j2 >> 1, par = j2 & 1;
        const bool fr = j < p.T;
        const int col = fr ? 0 : (j < p.T + p.H ? ybase + (j - p.T) : xbase + (j - p.T - p.H));
        float a = 0.f;
        for (int i0 = par; i0 < p.nparts; i0 += 16) {
            float v[8], wgt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 2 * u;
                const int ic = i < p.nparts ? i : p.nparts - 1;
                const int c = fr ? j - ff[ic] : col;
                const bool ok = i < p.nparts && (!fr || (c >= 0 && c < 8));
                v[u] = mg[ic * mstride + (ok ? c : 0)];
                wgt[u] = ok ? wp[ic] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fmaf(wgt[u], v[u], a);
        }
        a += __shfl_xor(a, 1, 64);
        if (par == 0) wrow[2 + p.nparts + j] = a;
    }
    for (int i = tid; i < p.nparts; i += 256) wrow[2 + i] = wp[i];
    if (tid == 0) {
        wrow[0] = M;
        wrow[1] = L;
        p.out_ml[2 * r] = M;
        p.out_ml[2 * r + 1] = L;
    }
}

__global__ __launch_bounds__(256) void combine_kernel(const float* ml, const float* acc, long ml_stride, long acc_stride,
                                                      int nsets, int rows, int E, float* ctx) {
    const int r = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= E) return;
    float M = -1.0e30f;
    for (int k = 0; k < nsets; ++k) M = fmaxf(M, ml[k * ml_stride + 2 * r]);
    float L = 0.f, a = 0.f;
    for (int k = 0; k < nsets; ++k) {
        const float w = expf(ml[k * ml_stride + 2 * r] - M);
        L = fmaf(w, ml[k * ml_stride + 2 * r + 1], L);
        a = fmaf(w, acc[k * acc_stride + (long)r * E + c], a);
    }
    ctx[(long)r * E + c] = a / L;
}

// ---------------------------------------------------------------------------------------------
// Release recipe, single query (direct mode): merge of the ring kernel's partial states FUSED with v_proj
// (reference projector.py:182,215 after folding: o_h = W_v,h ctx_h).  One workgroup per (head h, 64-channel slab):
//   M, L of the row; ctx_h[slab] = sum_p e^(m_p - M) acc_p[h, slab] / L        (the plain merge; pos-emb already inside acc)
//   po[slab, h*hd + j] = sum_{c in slab} W_v[h*hd + j, c] ctx_h[c]               (partial v_proj over the slab)
// The E/64 partial vectors po[slab, :] are summed (in slab order: deterministic, no atomics) by the consumer, the
// out_proj GEMV that rides in the first readout GEMM's launch.  Also emits the merged state (ml, ctx) for callers
// that want it (may be NULL).
struct MergeVprojParams {
    const float* part_m;
    const float* part_l;
    const float* part_acc;
    int nparts, rows_pad, E, hd;
    const uint16_t* wv;     // bf16 [E, E]
    float* po;              // [E / 64][E]
    float* out_ml;          // [R][2] or NULL
    float* out_ctx;         // [R][E] normalised, or NULL
};

__global__ __launch_bounds__(256) void merge_vproj_kernel(MergeVprojParams p) {
    __shared__ float wp[256];
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float cpart[16][64];
    __shared__ __attribute__((aligned(16))) float cx[64];
    const int slab = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    // the v_proj weights of this (head, slab) first: they depend on nothing in this kernel, and behind the two block reductions
    // below their (cold) load was a third dependent memory round trip at the end of the launch
    const int j = tid >> 1, half = tid & 1;
    u32x4 wreg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        wreg[q] = (j < p.hd) ? *reinterpret_cast<const u32x4*>(p.wv + (long)(h * p.hd + j) * p.E + slab * 64 + 32 * half + 8 * q) : u32x4{0, 0, 0, 0};
    // raw partial rows (independent of M): thread = (float4 column c4 of the slab, partial group pg of 16);
    // all of a thread's <= 16 loads are in flight together
    const int c4 = tid & 15, pg = tid >> 4;
    const float* base = p.part_acc + (long)h * p.E + slab * 64 + 4 * c4;
    const long pstride = (long)p.rows_pad * p.E;
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int i = pg + 16 * u;
        v[u] = (i < p.nparts) ? *reinterpret_cast<const float4*>(base + (long)i * pstride) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float mx = -1.0e30f;
    for (int i = tid; i < p.nparts; i += 256) mx = fmaxf(mx, p.part_m[(long)i * p.rows_pad + h]);
    const float M = block_reduce_max(mx, red);
    float ls = 0.f;
    for (int i = tid; i < 256; i += 256) {
        const float w = i < p.nparts ? expf(p.part_m[(long)i * p.rows_pad + h] - M) : 0.f;
        wp[i] = w;
        ls += i < p.nparts ? w * p.part_l[(long)i * p.rows_pad + h] : 0.f;
    }
    const float L = block_reduce_sum(ls, red);       // (barriers inside: wp[] is visible afterwards)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float w = wp[pg + 16 * u];
        a.x = fmaf(w, v[u].x, a.x); a.y = fmaf(w, v[u].y, a.y); a.z = fmaf(w, v[u].z, a.z); a.w = fmaf(w, v[u].w, a.w);
    }
    *reinterpret_cast<float4*>(&cpart[pg][4 * c4]) = a;
    __syncthreads();
    if (tid < 64) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) sum += cpart[g][tid];
        const float val = sum / L;
        cx[tid] = val;
        if (p.out_ctx) p.out_ctx[(long)h * p.E + slab * 64 + tid] = val;
    }
    if (tid == 0 && slab == 0 && p.out_ml) {
        p.out_ml[2 * h] = M;
        p.out_ml[2 * h + 1] = L;
    }
    __syncthreads();
    // partial v_proj: thread (j = tid >> 1, half = tid & 1) dots 32 channels of weight row h*hd + j
    float dot = 0.f;
    if (j < p.hd) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 g = wreg[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dot = fmaf(bf16lo_to_f32(g[i]), cx[32 * half + 8 * q + 2 * i], dot);
                dot = fmaf(bf16hi_to_f32(g[i]), cx[32 * half + 8 * q + 2 * i + 1], dot);
            }
        }
    }
    dot += __shfl_xor(dot, 1, 64);
    if (half == 0 && j < p.hd) p.po[(long)slab * p.E + h * p.hd + j] = dot;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same merge + v_proj with the slab sums taken HERE (round 4).  The E/64 partial vectors of the form above cost their
// consumer -- the GEMV role inside readout GEMM 1's launch, ~48 workgroups on the CUs the tile grid leaves idle -- 83 KB of
// cold reads per workgroup, and what ONE CU pulls from cold memory (~15-20 GB/s) is what bounds these small kernels: GEMM 1
// took 15.8 us with the 18 partials against 12.8 us with one vector (profiles/r04_*).  Here every workgroup adds its partial dot
// products into ONE vector o_fix [E] of 64-bit FIXED-POINT accumulators (value * 2^36, integer atomic adds: associative, so
// the sum is the same whatever order the workgroups arrive in -- float atomics would make the step's results vary from launch
// to launch).  o_fix must be zero on entry: hicom_fused_stream_fwd zeroes it (same stream, one launch earlier).
// Narrower slabs (32 channels, 36 x 9 = 324 workgroups, two or more per CU): the 9 MB of partial states spread evenly over
// the chip instead of 71 KB on each of 162 CUs.
template <int kMvSlab, bool F16>
__global__ __launch_bounds__(256) void merge_vproj_fixed_kernel(MergeVprojFixParams p) {
    __shared__ __attribute__((aligned(16))) char lds[mv_item_lds_bytes<kMvSlab>()];
    merge_vproj_fixed_item<kMvSlab, F16>(p, blockIdx.x, blockIdx.y, lds);
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_global_merge_fwd(const float* part_m, const float* part_l, const float* part_acc,
                                      int32_t nparts, int32_t rows, int32_t rows_pad, int32_t E,
                                      const float* scores, int64_t score_stride, int64_t N,
                                      int32_t H, int32_t W, const float* pe,
                                      int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                      float* scratch, float* out_ml, float* out_acc, int32_t normalize,
                                      void* stream) {
    HICOM_REQUIRE(part_m && part_l && part_acc && out_ml && out_acc, HICOM_EINVAL, "global_merge: NULL pointer");
    HICOM_REQUIRE(nparts > 0 && rows > 0 && rows <= rows_pad && E > 0, HICOM_EINVAL, "global_merge: bad shape");
    hipStream_t s = (hipStream_t)stream;
    int T = 0;
    if (pe) {
        HICOM_REQUIRE(scores && scratch && H > 0 && W > 0 && N > 0 && N % ((long)H * W) == 0, HICOM_EINVAL,
                      "global_merge: positional term needs scores, scratch and N %% (H*W) == 0");
        T = (int)(N / ((long)H * W));
        const size_t smem = ((size_t)H * W + H + 4) * 4;
        HICOM_REQUIRE(smem <= 60000, HICOM_EUNSUP, "global_merge: grid %dx%d too large for the marginal kernel", H, W);
        if (H <= 32 && W <= 32)
            hipLaunchKernelGGL(frame_marginals_wave_kernel, dim3((unsigned)(((long)rows * T + 3) / 4)), dim3(256), 0, s, scores,
                               (long)score_stride, rows, T, H, W, scratch);
        else
            hipLaunchKernelGGL(frame_marginals_kernel, dim3((unsigned)rows, (unsigned)T), dim3(256), smem, s, scores,
                               (long)score_stride, T, H, W, scratch);
    }
    MergeCtxParams p{part_m, part_l, part_acc, nparts, rows_pad, E, pe ? scratch : nullptr, pe,
                     T, H, W, t_index0, y_index0, x_index0, out_ml, out_acc, normalize};
    HICOM_REQUIRE(E % 4 == 0 && (!pe || H + W <= 1024), HICOM_EUNSUP, "global_merge: E %% 4 and H + W <= 1024");
    const size_t smem2 = ((size_t)nparts + (pe ? (size_t)2 * T + H + W + (size_t)kTC * (H + W + 2) : 0) + 16 * 64 + 4) * 4;
    HICOM_REQUIRE(smem2 <= 60000, HICOM_EUNSUP, "global_merge: too many partials/frames for one pass");
    if (pe && (long)nparts + T + H + W + 2 <= (long)min(kTC, T) * (H + W + 2)) {
        HICOM_LAUNCH(merge_ctx_kernel<1>, dim3((unsigned)rows), dim3(256), smem2, s, p);
        launch_merge_apply(p, rows, smem2, s);
    } else {
        HICOM_LAUNCH(merge_ctx_kernel<0>, dim3((unsigned)rows, (unsigned)((E + 31) / 32)), dim3(256), smem2, s, p);
    }
    return hicom_host::check_launch("global_merge");
}

// Merge behind hicom_global_stream_marg_fwd: no logit tensor, no per-frame marginal pass.  scratch: f32, >= rows * T * (H + W + 2)
// (the size hicom_global_merge_fwd asks for; only the head of each row's region is used).
extern "C" int hicom_global_merge_marg_fwd(const float* part_m, const float* part_l, const float* part_acc, const float* part_marg,
                                           int32_t nparts, int32_t rows, int32_t rows_pad, int32_t E, int64_t N,
                                           int32_t H, int32_t W, const float* pe,
                                           int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                           float* scratch, float* out_ml, float* out_acc, int32_t normalize, void* stream) {
    HICOM_REQUIRE(part_m && part_l && part_acc && part_marg && pe && scratch && out_ml && out_acc, HICOM_EINVAL, "global_merge_marg: NULL pointer");
    HICOM_REQUIRE(nparts > 0 && rows > 0 && rows <= rows_pad && E > 0 && E % 4 == 0 && H > 0 && W > 0 && H <= 64 && W <= 64 && N > 0 &&
                      N % ((long)H * W) == 0, HICOM_EINVAL, "global_merge_marg: bad shape");
    const int T = (int)(N / ((long)H * W));
    HICOM_REQUIRE((long)2 + nparts + T + H + W <= (long)T * (H + W + 2), HICOM_EUNSUP, "global_merge_marg: weight record exceeds the row's scratch region");
    hipStream_t s = (hipStream_t)stream;
    MergeCtxParams p{part_m, part_l, part_acc, nparts, rows_pad, E, scratch, pe, T, H, W, t_index0, y_index0, x_index0, out_ml, out_acc, normalize};
    const size_t smem1 = ((size_t)2 * nparts + 8) * 4;
    const size_t smem2 = ((size_t)nparts + (size_t)2 * T + H + W + (size_t)kTC * (H + W + 2) + 16 * 64 + 4) * 4;
    HICOM_REQUIRE(smem2 <= 60000, HICOM_EUNSUP, "global_merge_marg: too many partials/frames for one pass");
    HICOM_LAUNCH(merge_marg_weights_kernel, dim3((unsigned)rows), dim3(256), smem1, s, p, part_marg, hicom_global_stream_marg_width(H, W),
                 (int)((N + 15) / 16));
    launch_merge_apply(p, rows, smem2, s);
    return hicom_host::check_launch("global_merge_marg");
}

extern "C" int hicom_global_combine_fwd(const float* ml, const float* acc, int32_t nsets, int32_t rows,
                                        int32_t E, float* ctx, void* stream) {
    HICOM_REQUIRE(ml && acc && ctx, HICOM_EINVAL, "global_combine: NULL pointer");
    HICOM_REQUIRE(nsets > 0 && rows > 0 && E > 0, HICOM_EINVAL, "global_combine: bad shape");
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)rows, (unsigned)((E + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, ml, acc, (long)rows * 2, (long)rows * E, nsets, rows, E, ctx);
    return hicom_host::check_launch("global_combine");
}

extern "C" int hicom_global_combine_strided_fwd(const float* ml, const float* acc, int64_t set_stride, int32_t nsets,
                                                int32_t rows, int32_t E, float* ctx, void* stream) {
    HICOM_REQUIRE(ml && acc && ctx, HICOM_EINVAL, "global_combine_strided: NULL pointer");
    HICOM_REQUIRE(nsets > 0 && rows > 0 && E > 0 && set_stride > 0, HICOM_EINVAL, "global_combine_strided: bad shape");
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)rows, (unsigned)((E + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, ml, acc, (long)set_stride, (long)set_stride, nsets, rows, E, ctx);
    return hicom_host::check_launch("global_combine_strided");
}

extern "C" int hicom_merge_vproj_fixed_fwd(const float* part_m, const float* part_l, const void* part_acc, int32_t part_dt, int32_t nparts,
                                           int32_t rows, int32_t rows_pad, int32_t E, const void* w_v, int64_t* o_fix,
                                           float* out_ml, float* out_ctx, void* stream) {
    HICOM_REQUIRE(part_m && part_l && part_acc && w_v && o_fix, HICOM_EINVAL, "merge_vproj_fixed: NULL pointer");
    HICOM_REQUIRE(nparts > 0 && nparts <= 256 && rows > 0 && rows <= rows_pad && E > 0 && E % 64 == 0 && E % rows == 0 &&
                      E / rows <= 128 && ((uintptr_t)o_fix % 8 == 0) && ((uintptr_t)w_v % 16 == 0) && ((uintptr_t)part_acc % 16 == 0),
                  HICOM_EINVAL, "merge_vproj_fixed: bad shape (nparts <= 256, head dim <= 128, E %% 32) or alignment");
    MergeVprojFixParams p{part_m, part_l, part_acc, nparts, rows_pad, E, E / rows, (const uint16_t*)w_v, (long long*)o_fix, out_ml, out_ctx};
    HICOM_REQUIRE(part_dt == HICOM_DT_F32 || part_dt == HICOM_DT_F16, HICOM_EINVAL, "merge_vproj_fixed: part_dt");
    // (32-channel slabs -- 324 workgroups -- measured 5.9 us against 5.5 us for 64: the atomics, not the partial reads, grow)
    if (part_dt == HICOM_DT_F16) HICOM_LAUNCH((merge_vproj_fixed_kernel<64, true>), dim3((unsigned)(E / 64), (unsigned)rows), dim3(256), 0, (hipStream_t)stream, p);
    else HICOM_LAUNCH((merge_vproj_fixed_kernel<64, false>), dim3((unsigned)(E / 64), (unsigned)rows), dim3(256), 0, (hipStream_t)stream, p);
    return hicom_host::check_launch("merge_vproj_fixed");
}

extern "C" int hicom_merge_vproj_sets_fwd(const float* sets, int64_t set_stride, int32_t nsets, int32_t rows, int32_t E, const void* w_v, int64_t* o_fix,
                                          float* out_ml, float* out_ctx, void* stream) {
    HICOM_REQUIRE(sets && w_v && o_fix, HICOM_EINVAL, "merge_vproj_sets: NULL pointer");
    HICOM_REQUIRE(nsets > 0 && nsets <= 256 && rows > 0 && E > 0 && E % 64 == 0 && E % rows == 0 && E / rows <= 128 && set_stride >= 2 * rows + (int64_t)rows * E &&
                      set_stride % 2 == 0 && ((uintptr_t)sets % 8 == 0) && ((uintptr_t)o_fix % 8 == 0) && ((uintptr_t)w_v % 16 == 0),
                  HICOM_EINVAL, "merge_vproj_sets: bad shape (nsets <= 256, head dim <= 128, E %% 64, 8-byte aligned sets)");
    MergeVprojFixParams p{sets, sets + 1, sets + 2 * rows, nsets, rows, E, E / rows, (const uint16_t*)w_v, (long long*)o_fix, out_ml, out_ctx};
    p.ml_part = set_stride; p.ml_row = 2; p.acc_part = set_stride; p.acc_row = E;
    HICOM_LAUNCH((merge_vproj_fixed_kernel<64, false>), dim3((unsigned)(E / 64), (unsigned)rows), dim3(256), 0, (hipStream_t)stream, p);
    return hicom_host::check_launch("merge_vproj_sets");
}

extern "C" int hicom_merge_vproj_fwd(const float* part_m, const float* part_l, const float* part_acc, int32_t nparts,
                                     int32_t rows, int32_t rows_pad, int32_t E, const void* w_v, float* po,
                                     float* out_ml, float* out_ctx, void* stream) {
    HICOM_REQUIRE(part_m && part_l && part_acc && w_v && po, HICOM_EINVAL, "merge_vproj: NULL pointer");
    HICOM_REQUIRE(nparts > 0 && nparts <= 256 && rows > 0 && rows <= rows_pad && E > 0 && E % 64 == 0 && E % rows == 0 &&
                      E / rows <= 128 && (E / rows) % 1 == 0, HICOM_EINVAL, "merge_vproj: bad shape (nparts <= 256, head dim <= 128)");
    MergeVprojParams p{part_m, part_l, part_acc, nparts, rows_pad, E, E / rows, (const uint16_t*)w_v, po, out_ml, out_ctx};
    HICOM_LAUNCH(merge_vproj_kernel, dim3((unsigned)(E / 64), (unsigned)rows), dim3(256), 0, (hipStream_t)stream, p);
    return hicom_host::check_launch("merge_vproj");
}
