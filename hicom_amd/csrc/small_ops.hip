// Small per-call operators around the two streaming kernels: few-row linears (GEMV class),
// the k_proj fold, the hi/lo bf16 split and row scatter.  All are latency/HBM-bound on a few MB
// of weights; they use plain fp32 FMAs on wave64 with 16-byte coalesced weight reads.
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace hicom_host {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local void* g_stop_event = nullptr;
void set_stop_event(void* ev) { g_stop_event = ev; }
void* take_stop_event() {
    void* ev = g_stop_event;
    g_stop_event = nullptr;
    return ev;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return HICOM_ELAUNCH;
    }
    return HICOM_OK;
}
}  // namespace hicom_host

extern "C" int hicom_abi_version(void) { return HICOM_ABI_VERSION; }
extern "C" const char* hicom_last_error(void) { return hicom_host::g_err; }

namespace hicom {

// ---------------------------------------------------------------------------------------------
// y[m, n] = act(sum_k x[row(m,n), k] * w[n, k] + b[n]) + res[m, n]
// one wave per output column n, up to MR rows per workgroup pass; lanes stride K in 8-element
// chunks (16 B of bf16 weights / 32 B of f32), 64-lane shuffle reduction per row.
// ---------------------------------------------------------------------------------------------
struct LinearParams {
    const void* x;
    const void* w;
    const void* b;
    const void* res;
    float* y;
    int x_f32, w_f32, b_f32, res_flags;   // res_flags: bit0 = broadcast row 0, bit1 = res is bf16
    int M, N, K;
    int head_rows, head_dim;
    int act;
    // optional second destination: row m is replicated to rows row0 + m + k*M (k < reps) of a packed
    // [*, ldd] tensor of dtype dst (the 32 global rows of the output, projector.py:646,707)
    void* dst;
    int dst_f32, reps;
    long ldd, row0;
};

template <bool F32>
__device__ __forceinline__ void load8(const void* base, long off, float (&v)[8]) {
    if constexpr (F32) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
        const u32x4 g = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = bf16lo_to_f32(g[i]);
            v[2 * i + 1] = bf16hi_to_f32(g[i]);
        }
    }
}

template <bool XF32, bool WF32, int MR>
__global__ __launch_bounds__(256) void linear_rows_kernel(LinearParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    const int m0 = blockIdx.y * MR;
    if (n >= p.N) return;
    float acc[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[r] = 0.f;
    const int head = p.head_dim > 0 ? n / p.head_dim : 0;
    // x rows of this pass (rows past M re-read the last one and are dropped at the store): their loads are unconditional, so
    // all MR of them and the weight chunk are in flight together -- with a branch around each row the compiler kept the loads
    // serial and a 32-row call cost MR L2 round trips per K chunk (20 us at 32 x 1152 x 1152)
    long xoff[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const int m = m0 + r < p.M ? m0 + r : p.M - 1;
        xoff[r] = (p.head_dim > 0 ? (long)m * p.head_rows + head : (long)m) * p.K;
    }
#pragma unroll 2
    for (int k = lane * 8; k < p.K; k += 512) {
        float wv[8], xv[MR][8];
        load8<WF32>(p.w, (long)n * p.K + k, wv);
#pragma unroll
        for (int r = 0; r < MR; ++r) load8<XF32>(p.x, xoff[r] + k, xv[r]);
#pragma unroll
        for (int r = 0; r < MR; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[r] = fmaf(xv[r][i], wv[i], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const float s = wave_sum_fast(acc[r]);
        const int m = m0 + r;
        if (lane == 0 && m < p.M) {
            float v = s;
            if (p.b) v += p.b_f32 ? reinterpret_cast<const float*>(p.b)[n]
                                  : bf16_to_f32(reinterpret_cast<const uint16_t*>(p.b)[n]);
            if (p.act == HICOM_ACT_GELU) v = gelu_erf(v);
            if (p.res) {
                const long ri = ((p.res_flags & 1) ? 0 : (long)m * p.N) + n;
                v += (p.res_flags & 2) ? bf16_to_f32(reinterpret_cast<const uint16_t*>(p.res)[ri]) : reinterpret_cast<const float*>(p.res)[ri];
            }
            if (p.y) p.y[(long)m * p.N + n] = v;
            if (p.dst) {
                for (int k = 0; k < p.reps; ++k) {
                    const long o = (p.row0 + m + (long)k * p.M) * p.ldd + n;
                    if (p.dst_f32) reinterpret_cast<float*>(p.dst)[o] = v;
                    else reinterpret_cast<uint16_t*>(p.dst)[o] = f32_to_bf16(v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same linear for 2 .. 64 rows on the matrix cores: y[M <= 64, N] = act(x . w^T + b) + res, bf16 weights.
// The GEMV form above re-reads every weight row once per 8 rows and spends ~150 VALU ops per lane and K chunk on bf16
// unpacking and shuffle reductions: 13-19 us for a 32..64-row call at 1152 x 1152, and the guide-off / coarse / fine recipes
// make 6-12 such calls per forward.  Here a workgroup owns 16 output columns; its 4 waves take every fourth K step of 32
// (B fragment = 16 weight rows x 64 B straight from global memory, A fragments = the x rows, fp32 inputs split into bf16
// hi + lo: two MFMAs per block), the four partial tiles meet in LDS and wave i finishes row block i.
// ---------------------------------------------------------------------------------------------
template <bool XF32, int MB>
__global__ __launch_bounds__(256) void linear_mfma_kernel(LinearParams p) {
    __shared__ __attribute__((aligned(16))) float red[4][MB][256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int head = p.head_dim > 0 ? n0 / p.head_dim : 0;
    const uint16_t* wrow = reinterpret_cast<const uint16_t*>(p.w) + (long)(n0 + r16) * p.K + 8 * kg;
    long xoff[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = 16 * i + r16 < p.M ? 16 * i + r16 : p.M - 1;
        xoff[i] = (p.head_dim > 0 ? (long)m * p.head_rows + head : (long)m) * p.K + 8 * kg;
    }
    f32x4 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = p.K >> 5;
    // K steps whose loads are in flight together: all 9 of a wave's share at K = 1152 where the registers allow (one memory
    // round trip per call instead of three)
    constexpr int U = XF32 ? (MB > 2 ? 3 : 9) : 9;
    for (int base = wave; base < nks; base += 4 * U) {
        bf16x8 wf[U];
        float xa[U][MB][8];
        bf16x8 xb[U][MB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = base + 4 * u < nks ? base + 4 * u : nks - 1;       // (clamped: a repeated step is zeroed below)
            wf[u] = *reinterpret_cast<const bf16x8*>(wrow + 32 * ks);
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                if constexpr (XF32) load8<true>(p.x, xoff[i] + 32 * ks, xa[u][i]);
                else xb[u][i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const uint16_t*>(p.x) + xoff[i] + 32 * ks);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                    // every load above is issued before the first use below
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + 4 * u >= nks) wf[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};     // a clamped (repeated) step adds nothing; no branch
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                if constexpr (XF32) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        uint16_t h, l;
                        split_bf16(xa[u][i][e], h, l);
                        hi[e] = (short)h;
                        lo[e] = (short)l;
                    }
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, wf[u], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, wf[u], acc[i], 0, 0, 0);
                } else {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[u][i], wf[u], acc[i], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) *reinterpret_cast<f32x4*>(&red[wave][i][4 * lane]) = acc[i];
    __syncthreads();
    for (int i = wave; i < MB; i += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][i][4 * lane]);
        v += *reinterpret_cast<const f32x4*>(&red[1][i][4 * lane]);
        v += *reinterpret_cast<const f32x4*>(&red[2][i][4 * lane]);
        v += *reinterpret_cast<const f32x4*>(&red[3][i][4 * lane]);
        const int n = n0 + r16;
        float bias = 0.f;
        if (p.b) bias = p.b_f32 ? reinterpret_cast<const float*>(p.b)[n] : bf16_to_f32(reinterpret_cast<const uint16_t*>(p.b)[n]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = 16 * i + 4 * kg + q;
            if (m >= p.M) continue;
            float o = v[q] + bias;
            if (p.act == HICOM_ACT_GELU) o = gelu_erf(o);
            if (p.res) {
                const long ri = ((p.res_flags & 1) ? 0 : (long)m * p.N) + n;
                o += (p.res_flags & 2) ? bf16_to_f32(reinterpret_cast<const uint16_t*>(p.res)[ri]) : reinterpret_cast<const float*>(p.res)[ri];
            }
            if (p.y) p.y[(long)m * p.N + n] = o;
            if (p.dst) {
                for (int k = 0; k < p.reps; ++k) {
                    const long off = (p.row0 + m + (long)k * p.M) * p.ldd + n;
                    if (p.dst_f32) reinterpret_cast<float*>(p.dst)[off] = o;
                    else reinterpret_cast<uint16_t*>(p.dst)[off] = f32_to_bf16(o);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Query fold.  For head h (hd = E / nh rows of k_proj) and query q:
//   qt[q*nh + h, c]    = scale * sum_j w_k[h*hd + j, c] * qp[q, h*hd + j]          c in [0, E)
//   pos_a[q*nh + h, p] = scale * sum_j kpe[h*hd + j, p] * qp[q, h*hd + j]          p in [0, P)
// where kpe = w_k . PE^T is a weight-only constant cached by the caller, so the score-side
// positional table qt . PE^T needs no second pass over qt.  qt is emitted as fp32 (optional)
// and as the bf16 hi/lo pair the MFMA stream kernel consumes.
// 256 threads = 4 j-groups x 64 lanes; lane owns 2 adjacent w_k columns (one 4-byte load per j)
// or 1 kpe column; partial sums of the 4 j-groups meet in LDS.
// ---------------------------------------------------------------------------------------------
struct FoldParams {
    const float* qp;
    const uint16_t* wk;
    const float* kpe;
    int nq, nh, E, P;
    float scale;
    float* qt;
    uint16_t* qhi;
    uint16_t* qlo;
    float* pos_a;
    int pos_stride;
    int nbE;
    // optional: broadcast one bf16 row (the local query of the fused stream kernel) into rows
    // [fill_row0, fill_row0 + fill_rows) of qhi -- saves a separate launch on the critical path
    const uint16_t* fill;
    int fill_row0, fill_rows;
};

// 256 threads = 16 j-groups x 16 lanes; a workgroup owns 32 w_k columns (2 per lane, one 4-byte load
// per j) or 16 kpe columns of one head: hd/16 = 8 serial steps per thread, E/32 * nh workgroups.
template <int QB>
__global__ __launch_bounds__(256) void fold_query_kernel(FoldParams p) {
    const int tid = threadIdx.x, jg = tid >> 4, cl = tid & 15;
    const int h = blockIdx.y, q0 = blockIdx.z * QB;
    if (h == p.nh) {   // fill blocks
        for (int i = blockIdx.x * 256 + tid; i < p.fill_rows * p.E; i += gridDim.x * 256)
            p.qhi[(long)p.fill_row0 * p.E + i] = p.fill[i % p.E];
        return;
    }
    const int hd = p.E / p.nh, jn = (hd + 15) / 16;
    __shared__ float qs[QB][128];          // hd <= 128
    __shared__ float red[16][QB][32];
    for (int i = tid; i < QB * hd; i += 256) {
        const int q = i / hd, j = i - q * hd;
        qs[q][j] = (q0 + q < p.nq) ? p.qp[(long)(q0 + q) * p.E + h * hd + j] : 0.f;
    }
    __syncthreads();
    const bool is_w = (int)blockIdx.x < p.nbE;
    float a0[QB], a1[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) a0[q] = a1[q] = 0.f;
    const int j0 = jg * jn, j1 = min(hd, j0 + jn);
    if (is_w) {
        const int c = blockIdx.x * 32 + 2 * cl;
        if (c < p.E) {
#pragma unroll 8
            for (int j = j0; j < j1; ++j) {
                const uint32_t w2 = *reinterpret_cast<const uint32_t*>(p.wk + (long)(h * hd + j) * p.E + c);
                const float w0 = bf16lo_to_f32(w2), w1 = bf16hi_to_f32(w2);
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    a0[q] = fmaf(w0, qs[q][j], a0[q]);
                    a1[q] = fmaf(w1, qs[q][j], a1[q]);
                }
            }
        }
    } else {
        const int pc = (blockIdx.x - p.nbE) * 16 + cl;
        if (pc < p.P) {
#pragma unroll 8
            for (int j = j0; j < j1; ++j) {
                const float w0 = p.kpe[(long)(h * hd + j) * p.P + pc];
#pragma unroll
                for (int q = 0; q < QB; ++q) a0[q] = fmaf(w0, qs[q][j], a0[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        red[jg][q][2 * cl] = a0[q];
        red[jg][q][2 * cl + 1] = a1[q];
    }
    __syncthreads();
    for (int o = tid; o < QB * 32; o += 256) {
        const int q = o >> 5, i = o & 31;
        if (q0 + q >= p.nq) continue;
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += red[g][q][i];
        v *= p.scale;
        const long row = (long)(q0 + q) * p.nh + h;
        if (is_w) {
            const int c = blockIdx.x * 32 + i;
            if (c < p.E) {
                if (p.qt) p.qt[row * p.E + c] = v;
                if (p.qhi) {
                    uint16_t hi, lo;
                    split_bf16(v, hi, lo);
                    p.qhi[row * p.E + c] = hi;
                    p.qlo[row * p.E + c] = lo;
                }
            }
        } else if ((i & 1) == 0) {
            const int pc = (blockIdx.x - p.nbE) * 16 + (i >> 1);
            if (pc < p.P) p.pos_a[row * p.pos_stride + pc] = v;
        }
    }
}

__global__ __launch_bounds__(256) void split_bf16_kernel(const float* x, int rows, int rows_pad, int E,
                                                         uint16_t* hi, uint16_t* lo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows_pad * E) return;
    const long r = i / E;
    uint16_t h = 0, l = 0;
    if (r < rows) split_bf16(x[i], h, l);
    hi[i] = h;
    lo[i] = l;
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const void* src, int src_f32, int src_rows, int ncols,
                                                           void* dst, int dst_f32, long ldd, long row0, long row_step,
                                                           int nl_group, int count) {
    const int i = blockIdx.x;
    const long sr = i % src_rows, dr = row0 + (long)i * row_step + (nl_group > 0 ? i / nl_group : 0);
    for (int c = threadIdx.x; c < ncols; c += 256) {
        const float v = src_f32 ? reinterpret_cast<const float*>(src)[sr * ncols + c]
                                : bf16_to_f32(reinterpret_cast<const uint16_t*>(src)[sr * ncols + c]);
        if (dst_f32) reinterpret_cast<float*>(dst)[dr * ldd + c] = v;
        else reinterpret_cast<uint16_t*>(dst)[dr * ldd + c] = f32_to_bf16(v);
    }
}

// Rows of `nblocks` equal blocks that sit `block_stride` bytes apart (the per-rank segments of an all-gathered
// buffer) -> consecutive packed rows of the output (newline rows skipped every nl_group rows).  Same dtype on
// both sides: 16-byte copies, four rows per workgroup.
__global__ __launch_bounds__(256) void place_blocks_kernel(const char* src, int block_rows, long block_stride, int row_bytes,
                                                           char* dst, long ldd_bytes, long row0, int nl_group, int count) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= count) return;
    const int b = i / block_rows, r = i - b * block_rows;
    const char* s = src + (long)b * block_stride + (long)r * row_bytes;
    char* d = dst + (row0 + i + (nl_group > 0 ? i / nl_group : 0)) * ldd_bytes;
    for (int c = lane * 16; c < row_bytes; c += 64 * 16) *reinterpret_cast<u32x4*>(d + c) = *reinterpret_cast<const u32x4*>(s + c);
}

// clip-scale global stage: L2-normalise the projected queries (reference projector.py:185) and form the key-bias constants
__global__ __launch_bounds__(256) void clip_query_prep_kernel(float* qp, const uint16_t* bk, int nh, int E, float scale, float* c) {
    __shared__ float red[4];
    const int q = blockIdx.x, tid = threadIdx.x, hd = E / nh;
    float ss = 0.f;
    for (int i = tid; i < E; i += 256) { const float v = qp[(long)q * E + i]; ss = fmaf(v, v, ss); }
    ss = wave_sum_fast(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = tid; i < E; i += 256) qp[(long)q * E + i] *= inv;
    __syncthreads();
    const int lane = tid & 63;
    for (int h = tid >> 6; h < nh; h += 4) {
        float d = 0.f;
        if (bk)
            for (int j = lane; j < hd; j += 64) d = fmaf(qp[(long)q * E + h * hd + j], bf16_to_f32(bk[h * hd + j]), d);
        d = wave_sum_fast(d);
        if (lane == 0) c[q * nh + h] = scale * d;
    }
}

__global__ __launch_bounds__(256) void inv_norm_kernel(const float* ssq, int parts, long M, float* inv) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s = 0.f;
    for (int k = 0; k < parts; ++k) s += ssq[(long)k * M + m];
    inv[m] = 1.0f / sqrtf(s);
}

__global__ __launch_bounds__(256) void partials_sum_kernel(const float* parts, int nparts, long M, float* out) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += parts[(long)k * M + m];     // slice order: deterministic
    out[m] = s;
}

// Many partials of a short vector (column partials of the streaming backward kernels: 1024 x 1152): the loop above is one thread per
// output walking all partials one load at a time.  Here a workgroup owns 16 outputs and 16 lanes share each output's partials
// (k = g, g + 16, ...: 8 loads in flight per thread), combined in a fixed order through LDS: deterministic as well.
__global__ __launch_bounds__(256) void partials_sum_tall_kernel(const float* parts, int nparts, long M, float* out) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const long m = (long)blockIdx.x * 16 + c;
    float s = 0.f;
    if (m < M) {
        int k = g;
        for (; k + 16 * 7 < nparts; k += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = parts[(long)(k + 16 * u) * M + m];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nparts; k += 16) s += parts[(long)k * M + m];
    }
    red[g][c] = s;
    __syncthreads();
    if (threadIdx.x < 16 && m < M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
        out[m] = t;
    }
}

}  // namespace hicom

using namespace hicom;

// d frames_feature of the global stage (direct recipe: <= 16 folded rows).  A workgroup takes 64 tokens; thread t owns channels
// 4t .. 4t+3 with the 2 * rows coefficient rows (qt | dctx) of those channels in registers and walks the tokens: per token 2 * rows
// broadcast LDS reads of the coefficients (dS, p), 8 * rows FMAs, one 8-byte store.
template <int RMAX>
__global__ __launch_bounds__(320) void global_dx_kernel(const float* S, const float* dS, long stride, const float* ml, const float* qt, const float* dctx,
                                                          int rows, long N, int E, uint16_t* dx, int accumulate) {
    __shared__ float coef[64][2 * RMAX];
    const int tid = threadIdx.x;
    const long n0 = (long)blockIdx.x * 64;
    for (int idx = tid; idx < 64 * rows; idx += blockDim.x) {
        const int r = idx / 64, t = idx - r * 64;
        const long n = n0 + t;
        float a = 0.f, b = 0.f;
        if (n < N) {
            a = dS[(long)r * stride + n];
            b = expf(S[(long)r * stride + n] - ml[2 * r]) / ml[2 * r + 1];
        }
        coef[t][r] = a;
        coef[t][RMAX + r] = b;
    }
    const int c4 = 4 * tid;
    float4 wq[RMAX], wd[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        wq[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        wd[r] = wq[r];
        if (r < rows && c4 < E) {
            wq[r] = *reinterpret_cast<const float4*>(qt + (long)r * E + c4);
            wd[r] = *reinterpret_cast<const float4*>(dctx + (long)r * E + c4);
        }
    }
    __syncthreads();
    if (c4 >= E) return;
    for (int t = 0; t < 64 && n0 + t < N; ++t) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        uint16_t* o = dx + (n0 + t) * (long)E + c4;
        if (accumulate) {
            const uint2 old = *reinterpret_cast<const uint2*>(o);
            a = make_float4(bf16lo_to_f32(old.x), bf16hi_to_f32(old.x), bf16lo_to_f32(old.y), bf16hi_to_f32(old.y));
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const float ca = coef[t][r], cb = coef[t][RMAX + r];
            a.x = fmaf(ca, wq[r].x, a.x); a.y = fmaf(ca, wq[r].y, a.y); a.z = fmaf(ca, wq[r].z, a.z); a.w = fmaf(ca, wq[r].w, a.w);
            a.x = fmaf(cb, wd[r].x, a.x); a.y = fmaf(cb, wd[r].y, a.y); a.z = fmaf(cb, wd[r].z, a.z); a.w = fmaf(cb, wd[r].w, a.w);
        }
        *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)f32_to_bf16(a.x) | ((unsigned)f32_to_bf16(a.y) << 16),
                                                  (unsigned)f32_to_bf16(a.z) | ((unsigned)f32_to_bf16(a.w) << 16));
    }
}

extern "C" int hicom_global_dx_fwd(const float* S, const float* dS, int64_t score_stride, const float* ml, const float* qt, const float* dctx,
                                   int32_t rows, int64_t N, int32_t E, void* dx, int32_t accumulate, void* stream) {
    HICOM_REQUIRE(S && dS && ml && qt && dctx && dx, HICOM_EINVAL, "global_dx: NULL pointer");
    HICOM_REQUIRE(rows > 0 && rows <= 16 && N > 0 && E > 0 && E % 4 == 0 && E <= 1280 && score_stride >= N && ((uintptr_t)dx % 8 == 0) &&
                      ((uintptr_t)qt % 16 == 0) && ((uintptr_t)dctx % 16 == 0), HICOM_EUNSUP,
                  "global_dx: rows=%d (<= 16: the direct recipe's folded rows), E=%d (<= 1280, %% 4)", rows, E);
    const unsigned grid = (unsigned)((N + 63) / 64);
    if (rows <= 9) hipLaunchKernelGGL(global_dx_kernel<9>, dim3(grid), dim3(320), 0, (hipStream_t)stream, S, dS, (long)score_stride, ml, qt, dctx, rows, (long)N, E, (uint16_t*)dx, accumulate);
    else hipLaunchKernelGGL(global_dx_kernel<16>, dim3(grid), dim3(320), 0, (hipStream_t)stream, S, dS, (long)score_stride, ml, qt, dctx, rows, (long)N, E, (uint16_t*)dx, accumulate);
    return hicom_host::check_launch("global_dx");
}

// 16-bit <-> 16-bit cast of a contiguous tensor: fp16 -> bf16 (round to nearest even: the boundary accepts fp16 modules and inputs -- the
// reference's inference default, inference_video_mcqa_videomme.py:323 -- and runs them on the bf16 kernels) and bf16 -> fp16 (saturating:
// the result of such a call).  Eight elements per thread, 16-byte accesses.
__global__ __launch_bounds__(256) void cast16_kernel(const uint16_t* src, uint16_t* dst, long n8, long n, int to_bf16) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    auto conv = [&](uint16_t h) -> uint16_t {
        if (to_bf16) return f32_to_bf16((float)__builtin_bit_cast(_Float16, h));
        const float v = fminf(fmaxf(bf16_to_f32(h), -65504.f), 65504.f);
        const _Float16 o = (_Float16)v;
        return __builtin_bit_cast(uint16_t, o);
    };
    if (8 * i + 8 <= n) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + 8 * i);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned u = v[q];
            o[q] = (unsigned)conv((uint16_t)(u & 0xFFFFu)) | ((unsigned)conv((uint16_t)(u >> 16)) << 16);
        }
        *reinterpret_cast<u32x4*>(dst + 8 * i) = o;
    } else {
        for (long k = 8 * i; k < n; ++k) dst[k] = conv(src[k]);
    }
}

extern "C" int hicom_cast16_fwd(const void* src, int32_t src_dt, void* dst, int32_t dst_dt, int64_t n, void* stream) {
    HICOM_REQUIRE(src && dst && n > 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), HICOM_EINVAL, "cast16: bad arguments (16-byte aligned tensors)");
    HICOM_REQUIRE((src_dt == HICOM_DT_F16 && dst_dt == HICOM_DT_BF16) || (src_dt == HICOM_DT_BF16 && dst_dt == HICOM_DT_F16), HICOM_EINVAL,
                  "cast16: fp16 -> bf16 or bf16 -> fp16");
    const long n8 = (n + 7) / 8;
    hipLaunchKernelGGL(cast16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst, n8,
                       (long)n, dst_dt == HICOM_DT_BF16 ? 1 : 0);
    return hicom_host::check_launch("cast16");
}

extern "C" int hicom_partials_sum_fwd(const float* parts, int32_t nparts, int64_t M, float* out, void* stream) {
    HICOM_REQUIRE(parts && out && nparts > 0 && M > 0, HICOM_EINVAL, "partials_sum: bad arguments");
    if (nparts >= 64 && M <= 65536)
        hipLaunchKernelGGL(partials_sum_tall_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, (hipStream_t)stream, parts, nparts, (long)M, out);
    else
        hipLaunchKernelGGL(partials_sum_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, parts, nparts, (long)M, out);
    return hicom_host::check_launch("partials_sum");
}

extern "C" int hicom_clip_query_prep_fwd(float* qp, const void* b_k, int32_t nq, int32_t nh, int32_t E, float scale, float* c, void* stream) {
    HICOM_REQUIRE(qp && c && nq > 0 && nh > 0 && E > 0 && E % nh == 0, HICOM_EINVAL, "clip_query_prep: bad arguments");
    hipLaunchKernelGGL(clip_query_prep_kernel, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, qp, (const uint16_t*)b_k, nh, E, scale, c);
    return hicom_host::check_launch("clip_query_prep");
}

extern "C" int hicom_inv_norm_fwd(const float* ssq, int32_t parts, int64_t M, float* inv, void* stream) {
    HICOM_REQUIRE(ssq && inv && parts > 0 && M > 0, HICOM_EINVAL, "inv_norm: bad arguments");
    hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ssq, parts, (long)M, inv);
    return hicom_host::check_launch("inv_norm");
}

static int launch_linear(const LinearParams& p, void* stream) {
    const int M = p.M, N = p.N;
    hipStream_t s = (hipStream_t)stream;
#define HICOM_LAUNCH_LINEAR(MR)                                                                                        \
    do {                                                                                                               \
        dim3 grid((unsigned)((N + 3) / 4), (unsigned)((M + (MR)-1) / (MR)));                                            \
        if (p.x_f32 && p.w_f32) HICOM_LAUNCH((linear_rows_kernel<true, true, MR>), grid, dim3(256), 0, s, p);           \
        else if (p.x_f32) HICOM_LAUNCH((linear_rows_kernel<true, false, MR>), grid, dim3(256), 0, s, p);                \
        else if (p.w_f32) HICOM_LAUNCH((linear_rows_kernel<false, true, MR>), grid, dim3(256), 0, s, p);                \
        else HICOM_LAUNCH((linear_rows_kernel<false, false, MR>), grid, dim3(256), 0, s, p);                            \
    } while (0)
    const bool mfma = M >= 2 && M <= 64 && !p.w_f32 && p.K % 32 == 0 && N % 16 == 0 && (p.head_dim == 0 || p.head_dim % 16 == 0) &&
                      (uintptr_t)p.x % 16 == 0 && (uintptr_t)p.w % 16 == 0;
    if (mfma) {
        const dim3 grid((unsigned)(N / 16));
        const int mb = (M + 15) / 16;
#define HICOM_LAUNCH_MFMA(MB)                                                                                           \
    do {                                                                                                               \
        if (p.x_f32) HICOM_LAUNCH((linear_mfma_kernel<true, MB>), grid, dim3(256), 0, s, p);                            \
        else HICOM_LAUNCH((linear_mfma_kernel<false, MB>), grid, dim3(256), 0, s, p);                                   \
    } while (0)
        if (mb == 1) HICOM_LAUNCH_MFMA(1);
        else if (mb == 2) HICOM_LAUNCH_MFMA(2);
        else if (mb == 3) HICOM_LAUNCH_MFMA(3);
        else HICOM_LAUNCH_MFMA(4);
#undef HICOM_LAUNCH_MFMA
    } else if (M == 1) HICOM_LAUNCH_LINEAR(1);     // GEMV: one shuffle reduction per column instead of eight
    else HICOM_LAUNCH_LINEAR(8);
#undef HICOM_LAUNCH_LINEAR
    return hicom_host::check_launch("linear");
}

extern "C" int hicom_linear_fwd(const void* x, int32_t x_dt, const void* w, int32_t w_dt,
                                const void* b, int32_t b_dt, const void* res, int32_t res_flags,
                                int32_t M, int32_t N, int32_t K, int32_t head_rows, int32_t head_dim,
                                int32_t act, float* y, void* stream) {
    HICOM_REQUIRE(x && w && y, HICOM_EINVAL, "linear: NULL pointer");
    HICOM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0, HICOM_EINVAL, "linear: bad shape M=%d N=%d K=%d (K %% 8)", M, N, K);
    HICOM_REQUIRE(head_dim == 0 || (head_dim > 0 && head_rows > 0), HICOM_EINVAL, "linear: head mode");
    LinearParams p{x, w, b, res, y, x_dt == HICOM_DT_F32, w_dt == HICOM_DT_F32, b_dt == HICOM_DT_F32, res_flags,
                   M, N, K, head_rows, head_dim, act, nullptr, 0, 0, 0, 0};
    return launch_linear(p, stream);
}

extern "C" int hicom_linear_to_rows_fwd(const void* x, int32_t x_dt, const void* w, int32_t w_dt,
                                        const void* b, int32_t b_dt, int32_t M, int32_t N, int32_t K, int32_t act,
                                        void* dst, int32_t dst_dt, int64_t ldd, int64_t row0, int32_t n_rows,
                                        void* stream) {
    HICOM_REQUIRE(x && w && dst, HICOM_EINVAL, "linear_to_rows: NULL pointer");
    HICOM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && n_rows >= M && n_rows % M == 0 && ldd >= N, HICOM_EINVAL,
                  "linear_to_rows: bad shape M=%d N=%d K=%d rows=%d", M, N, K, n_rows);
    LinearParams p{x, w, b, nullptr, nullptr, x_dt == HICOM_DT_F32, w_dt == HICOM_DT_F32, b_dt == HICOM_DT_F32, 0,
                   M, N, K, 0, 0, act, dst, dst_dt == HICOM_DT_F32, n_rows / M, (long)ldd, (long)row0};
    return launch_linear(p, stream);
}




static int launch_fold(const float* qp, const void* w_k, const float* kpe, int nq, int nh, int E, int P, float scale,
                       float* qt, void* hi, void* lo, float* pos_a, int pos_stride, const void* fill, int fill_row0,
                       int fill_rows, void* stream) {
    FoldParams p{qp, (const uint16_t*)w_k, kpe, nq, nh, E, kpe ? P : 0, scale, qt, (uint16_t*)hi, (uint16_t*)lo,
                 pos_a, pos_stride, (E + 31) / 32, (const uint16_t*)fill, fill_row0, fill ? fill_rows : 0};
    const unsigned gx = (unsigned)(p.nbE + (p.P + 15) / 16), gy = (unsigned)(nh + (p.fill_rows > 0 ? 1 : 0));
    if (nq == 1) HICOM_LAUNCH(fold_query_kernel<1>, dim3(gx, gy, 1), dim3(256), 0, (hipStream_t)stream, p);
    else HICOM_LAUNCH(fold_query_kernel<8>, dim3(gx, gy, (unsigned)((nq + 7) / 8)), dim3(256), 0, (hipStream_t)stream, p);
    return hicom_host::check_launch("fold_query");
}

extern "C" int hicom_fold_query_fwd(const float* qp, const void* w_k, int32_t nq, int32_t nh, int32_t E,
                                    float scale, float* qt, void* stream) {
    HICOM_REQUIRE(qp && w_k && qt, HICOM_EINVAL, "fold_query: NULL pointer");
    HICOM_REQUIRE(nq > 0 && nh > 0 && E > 0 && E % nh == 0 && E / nh <= 128 && E % 2 == 0, HICOM_EINVAL, "fold_query: bad shape");
    return launch_fold(qp, w_k, nullptr, nq, nh, E, 0, scale, qt, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, stream);
}

extern "C" int hicom_fold_query_split_fwd(const float* qp, const void* w_k, const float* kpe, int32_t nq, int32_t nh,
                                          int32_t E, int32_t P, float scale, void* qt_hi, void* qt_lo,
                                          float* pos_a, int32_t pos_stride, const void* fill_row, int32_t fill_row0,
                                          int32_t fill_rows, void* stream) {
    HICOM_REQUIRE(qp && w_k && qt_hi && qt_lo, HICOM_EINVAL, "fold_query_split: NULL pointer");
    HICOM_REQUIRE(nq > 0 && nh > 0 && E > 0 && E % nh == 0 && E / nh <= 128 && E % 2 == 0, HICOM_EINVAL, "fold_query_split: bad shape");
    HICOM_REQUIRE(!kpe || (pos_a && P > 0 && pos_stride >= P), HICOM_EINVAL, "fold_query_split: positional outputs");
    HICOM_REQUIRE(!fill_row || (fill_row0 >= nq * nh && fill_rows > 0), HICOM_EINVAL, "fold_query_split: fill rows overlap the queries");
    return launch_fold(qp, w_k, kpe, nq, nh, E, P, scale, nullptr, qt_hi, qt_lo, pos_a, pos_stride, fill_row, fill_row0,
                       fill_rows, stream);
}

extern "C" int hicom_split_bf16_fwd(const float* x, int32_t rows, int32_t rows_pad, int32_t E,
                                    void* hi, void* lo, void* stream) {
    HICOM_REQUIRE(x && hi && lo, HICOM_EINVAL, "split_bf16: NULL pointer");
    HICOM_REQUIRE(rows > 0 && rows_pad >= rows && E > 0, HICOM_EINVAL, "split_bf16: bad shape");
    const long n = (long)rows_pad * E;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       rows_pad, E, (uint16_t*)hi, (uint16_t*)lo);
    return hicom_host::check_launch("split_bf16");
}

extern "C" int hicom_scatter_rows_fwd(const void* src, int32_t src_dt, int32_t src_rows, int32_t ncols,
                                      void* dst, int32_t dst_dt, int64_t ldd, int64_t row0, int64_t row_step,
                                      int32_t nl_group, int32_t count, void* stream) {
    HICOM_REQUIRE(src && dst, HICOM_EINVAL, "scatter_rows: NULL pointer");
    HICOM_REQUIRE(src_rows > 0 && ncols > 0 && count >= 0 && ldd >= ncols && nl_group >= 0, HICOM_EINVAL, "scatter_rows: bad shape");
    if (count == 0) return HICOM_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)count), dim3(256), 0, (hipStream_t)stream, src,
                       src_dt == HICOM_DT_F32, src_rows, ncols, dst, dst_dt == HICOM_DT_F32, (long)ldd, (long)row0,
                       (long)row_step, nl_group, count);
    return hicom_host::check_launch("scatter_rows");
}

extern "C" int hicom_place_blocks_fwd(const void* src, int32_t block_rows, int32_t nblocks, int64_t block_stride_bytes,
                                      int32_t row_bytes, void* dst, int64_t ldd_bytes, int64_t row0, int32_t nl_group,
                                      void* stream) {
    HICOM_REQUIRE(src && dst, HICOM_EINVAL, "place_blocks: NULL pointer");
    HICOM_REQUIRE(block_rows > 0 && nblocks > 0 && row_bytes > 0 && row_bytes % 16 == 0 && ldd_bytes >= row_bytes && ldd_bytes % 16 == 0 &&
                      block_stride_bytes % 16 == 0 && nl_group >= 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0),
                  HICOM_EINVAL, "place_blocks: bad shape / alignment");
    const int count = block_rows * nblocks;
    HICOM_LAUNCH(place_blocks_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                       block_rows, (long)block_stride_bytes, row_bytes, (char*)dst, (long)ldd_bytes, (long)row0, nl_group, count);
    return hicom_host::check_launch("place_blocks");
}
