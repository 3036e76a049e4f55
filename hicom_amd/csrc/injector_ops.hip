// Row-wise operators of the instruction injector and the q/k/v adaptors
// (reference projector.py:315-397 GuideInjector, :431-457 / :533-541 adaptors):
//
//   hicom_row_ln_fwd   out = (1 - alpha) * src + alpha * (LayerNorm(x * (1 + mul) + add) * gamma + beta)
//       coarse   : x = visual query, mul/add = FiLM scale/shift (one broadcast row), alpha = 1      (:369-372)
//       fine     : x = query, add = attention output per row, alpha = 1                             (:392)
//       adapt_*  : x = proj(src), src = un-adapted tensor, alpha = learned scalar                   (:365,:533-541)
//   hicom_small_mha_fwd  per-row multi-head attention over a short key list (the L = 64 text tokens
//       of "fine" injection, :391): projected q [M,E], k/v [L,E] -> [M,E]
//
// One wave per row (E <= 4096), values kept in registers; LayerNorm statistics in fp32, eps = 1e-6.
#include "common.hpp"

namespace hicom {

struct RowLnParams {
    const void* x; int x_dt; long x_stride;
    const float* mul; long mul_stride;     // NULL or [*,E]; stride 0 = broadcast row
    const float* add; long add_stride;
    const void* gamma; const void* beta; int gb_dt;
    const void* src; int src_dt; long src_stride;   // blend source (NULL => alpha must be 1)
    const void* alpha_ptr; int alpha_dt;             // device scalar or NULL (=> alpha = 1)
    float eps;
    void* out; int out_dt; long out_stride;
    int M, E;
};

__device__ __forceinline__ float ld(const void* p, int dt, long i) {
    return dt == HICOM_DT_F32 ? reinterpret_cast<const float*>(p)[i] : bf16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
}

constexpr int kMaxPerLane = 64;   // E <= 4096

// NPL = channels per lane, compile-time so the row stays in registers (18 for E = 1152, 12 for 768)
template <int NPL>
__global__ __launch_bounds__(256) void row_ln_kernel(RowLnParams p) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    constexpr int n = NPL;
    float v[NPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const int c = lane + 64 * i;
        float t = 0.f;
        if (c < p.E) {
            t = ld(p.x, p.x_dt, m * p.x_stride + c);
            if (p.mul) t *= 1.0f + p.mul[m * p.mul_stride + c];
            if (p.add) t += p.add[m * p.add_stride + c];
        }
        v[i] = t;
        sum += t;
    }
    const float mean = wave_sum_fast(sum) / (float)p.E;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const int c = lane + 64 * i;
        const float d = c < p.E ? v[i] - mean : 0.f;
        var += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum_fast(var) / (float)p.E + p.eps);
    const float alpha = p.alpha_ptr ? ld(p.alpha_ptr, p.alpha_dt, 0) : 1.0f;
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const int c = lane + 64 * i;
        if (c >= p.E) continue;
        float y = (v[i] - mean) * rstd * ld(p.gamma, p.gb_dt, c) + ld(p.beta, p.gb_dt, c);
        if (p.src) y = (1.0f - alpha) * ld(p.src, p.src_dt, m * p.src_stride + c) + alpha * y;
        if (p.out_dt == HICOM_DT_F32) reinterpret_cast<float*>(p.out)[m * p.out_stride + c] = y;
        else reinterpret_cast<uint16_t*>(p.out)[m * p.out_stride + c] = f32_to_bf16(y);
    }
}

// 8-wide form (E % 8 == 0, E <= 1536, every row base 16-byte aligned): a lane owns up to three chunks of 8 consecutive
// channels, every operand comes in 16- / 32-byte loads issued together.  The scalar form above makes ~54 dependent 4-byte
// loads per lane for a FiLM row (29 us for 1 296 x 1 152).
__device__ __forceinline__ void ld8v(const void* base, int dt, long off, float (&v)[8]) {
    if (dt == HICOM_DT_F32) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
        const u32x4 g = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16lo_to_f32(g[i]); v[2 * i + 1] = bf16hi_to_f32(g[i]); }
    }
}

__global__ __launch_bounds__(256) void row_ln8_kernel(RowLnParams p) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int nch = p.E >> 3;
    float v[3][8], mu[3][8], ad[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ch = lane + 64 * c < nch ? lane + 64 * c : nch - 1;       // (clamped: the extra chunk is masked below)
        ld8v(p.x, p.x_dt, m * p.x_stride + 8 * ch, v[c]);
        if (p.mul) ld8v(p.mul, HICOM_DT_F32, m * p.mul_stride + 8 * ch, mu[c]);
        if (p.add) ld8v(p.add, HICOM_DT_F32, m * p.add_stride + 8 * ch, ad[c]);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bool in = lane + 64 * c < nch;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = v[c][i];
            if (p.mul) t *= 1.0f + mu[c][i];
            if (p.add) t += ad[c][i];
            v[c][i] = in ? t : 0.f;
            sum += v[c][i];
        }
    }
    const float mean = wave_sum_fast(sum) / (float)p.E;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (lane + 64 * c < nch)
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; var = fmaf(d, d, var); }
    const float rstd = 1.0f / sqrtf(wave_sum_fast(var) / (float)p.E + p.eps);
    const float alpha = p.alpha_ptr ? ld(p.alpha_ptr, p.alpha_dt, 0) : 1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ch = lane + 64 * c;
        if (ch >= nch) continue;
        float g[8], b[8], y[8];
        ld8v(p.gamma, p.gb_dt, 8 * ch, g);
        ld8v(p.beta, p.gb_dt, 8 * ch, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (v[c][i] - mean) * rstd * g[i] + b[i];
        if (p.src) {
            float sv[8];
            ld8v(p.src, p.src_dt, m * p.src_stride + 8 * ch, sv);
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = (1.0f - alpha) * sv[i] + alpha * y[i];
        }
        if (p.out_dt == HICOM_DT_F32) {
            float* o = reinterpret_cast<float*>(p.out) + m * p.out_stride + 8 * ch;
            *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
        } else {
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = f32_to_bf16(y[2 * i]) | ((uint32_t)f32_to_bf16(y[2 * i + 1]) << 16);
            *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + m * p.out_stride + 8 * ch) = o;
        }
    }
}

// ---- small MHA: one wave per (row, head); lane = key token (L <= 64) for the scores, lane = channel
// pair for the weighted sum; fp32 softmax (reference projector.py:197-215 with scale = hd^-1/2).
__global__ __launch_bounds__(256) void small_mha_kernel(const float* q, const float* k, const float* v, int M, int L,
                                                        int nh, int hd, float scale, float* out) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long)M * nh) return;
    const long m = wid / nh;
    const int h = (int)(wid - m * nh), E = nh * hd;
    const float* qh = q + m * E + h * hd;
    float s = -1.0e30f;
    if (lane < L) {
        const float* kh = k + (long)lane * E + h * hd;
        float d = 0.f;
        for (int c = 0; c < hd; ++c) d = fmaf(qh[c], kh[c], d);
        s = d * scale;
    }
    const float mx = wave_max_fast(s);
    const float e = lane < L ? expf(s - mx) : 0.f;
    const float pr = e / wave_sum_fast(e);
    for (int c0 = 0; c0 < hd; c0 += 64) {
        const int c = c0 + lane;
        float a = 0.f;
        for (int t = 0; t < L; ++t) {
            const float pt = __shfl(pr, t, 64);
            if (c < hd) a = fmaf(pt, v[(long)t * E + h * hd + c], a);
        }
        if (c < hd) out[m * E + h * hd + c] = a;
    }
}

// ---- small MHA, LDS form (head dim 64 / 96 / 128): a workgroup owns one head and a block of query rows, stages that head's K and V
// slices ([L <= 64][hd] fp32, 64 KB) in LDS once and serves its rows from there.  Scores: lane = key token, K rows padded to
// hd + 4 words; weighted sum: lane = channel pair, the weight of token t by v_readlane.  The form above (one wave per
// (row, head), K and V read from L2 with a 4.6-KB stride between lanes) took 208 us for the 1 296 x 9 x 64 problem of the
// fine-grained injector.
constexpr int kMhaRows = 32;          // query rows per workgroup (4 waves x 8)

template <int hd>
__global__ __launch_bounds__(256) void small_mha_lds_kernel(const float* q, const float* k, const float* v, int M, int L,
                                                            int nh, float scale, float* out) {
    static_assert(hd % 4 == 0 && hd <= 128, "head dim");
    extern __shared__ __attribute__((aligned(16))) float msm[];
    constexpr int KP = hd + 4;                // K row pitch: 16-byte aligned rows; lane t's chunk starts at bank 4 t + c (mod 64):
                                              // the 16 lanes of a ds_read_b128 pass cover all 64 banks
    float* Ks = msm;                          // [64][KP]
    float* Vs = msm + 64 * KP;                // [64][hd]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, E = nh * hd;
    {
        // float4 staging, every load of a thread in flight together (hd % 4 == 0: host-checked): 64 * hd / 4 / 256 <= 8 per matrix
        const int q4 = hd >> 2;                               // float4 per row
        float4 kv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u, t = i / q4, c = 4 * (i - t * q4);
            const bool in = i < 64 * q4 && t < L;
            kv[u] = in ? *reinterpret_cast<const float4*>(k + (long)t * E + h * hd + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            vv[u] = in ? *reinterpret_cast<const float4*>(v + (long)t * E + h * hd + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u, t = i / q4, c = 4 * (i - t * q4);
            if (i < 64 * q4) {
                *reinterpret_cast<float4*>(Ks + t * KP + c) = kv[u];
                *reinterpret_cast<float4*>(Vs + t * hd + c) = vv[u];
            }
        }
    }
    __syncthreads();
    // Each wave takes its 8 rows TOGETHER: one K row chunk from LDS (ds_read_b128, lane = key token) serves 8 dot products,
    // the 8 query rows come by wave-uniform (scalar) loads; one V element pair serves 8 weighted sums.  One row at a time the
    // kernel was bound by LDS reads (384 wave-level reads per row; 35 us for 1 296 rows x 9 heads).
    const int mb = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * kMhaRows + wave * 8));
    if (mb >= M) return;
    const float* krow = Ks + lane * KP;
    const float* __restrict__ qrow[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) qrow[r] = q + (long)(mb + r < M ? mb + r : M - 1) * E + h * hd;
    float d[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = 0.f;
#pragma unroll 4
    for (int c = 0; c < hd; c += 4) {
        const f32x4 kq = *reinterpret_cast<const f32x4*>(krow + c);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f32x4 qq = *reinterpret_cast<const f32x4*>(qrow[r] + c);     // same address in every lane
            d[r] = fmaf(qq[0], kq[0], fmaf(qq[1], kq[1], fmaf(qq[2], kq[2], fmaf(qq[3], kq[3], d[r]))));
        }
    }
    float pr[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float sc = lane < L ? d[r] * scale : -1.0e30f;
        const float mx = wave_max_fast(sc);
        const float e = lane < L ? expf(sc - mx) : 0.f;
        pr[r] = e / wave_sum_fast(e);
    }
    float a0[8], a1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) a0[r] = a1[r] = 0.f;
    const bool c0 = lane < hd, c1 = lane + 64 < hd;
    const float* v0 = Vs + (c0 ? lane : 0);
    const float* v1 = Vs + (c1 ? lane + 64 : 0);
#pragma unroll 8
    for (int t = 0; t < 64; ++t) {
        const float x0 = v0[t * hd], x1 = hd > 64 ? v1[t * hd] : 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float pt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pr[r]), t));
            a0[r] = fmaf(pt, x0, a0[r]);
            if (hd > 64) a1[r] = fmaf(pt, x1, a1[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const long m = mb + r;
        if (m >= M) break;
        if (c0) out[m * E + h * hd + lane] = a0[r];
        if (c1) out[m * E + h * hd + lane + 64] = a1[r];
    }
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_row_ln_fwd(const void* x, int32_t x_dt, int64_t x_stride,
                                const float* mul, int64_t mul_stride, const float* add, int64_t add_stride,
                                const void* gamma, const void* beta, int32_t gb_dt,
                                const void* src, int32_t src_dt, int64_t src_stride,
                                const void* alpha, int32_t alpha_dt, float eps,
                                void* out, int32_t out_dt, int64_t out_stride, int32_t M, int32_t E, void* stream) {
    HICOM_REQUIRE(x && gamma && beta && out, HICOM_EINVAL, "row_ln: NULL pointer");
    HICOM_REQUIRE(M > 0 && E > 0 && E <= 64 * kMaxPerLane, HICOM_EINVAL, "row_ln: bad shape M=%d E=%d", M, E);
    HICOM_REQUIRE(src || !alpha, HICOM_EINVAL, "row_ln: alpha without a blend source");
    RowLnParams p{x, x_dt, (long)x_stride, mul, (long)mul_stride, add, (long)add_stride, gamma, beta, gb_dt,
                  src, src_dt, (long)src_stride, alpha, alpha_dt, eps, out, out_dt, (long)out_stride, M, E};
    const dim3 grid((unsigned)((M + 3) / 4));
    auto al16 = [](const void* ptr, int64_t stride, int dt) {
        const int es = dt == HICOM_DT_F32 ? 4 : 2;
        return !ptr || ((uintptr_t)ptr % 16 == 0 && (stride * es) % 16 == 0);
    };
    if (E % 8 == 0 && E <= 1536 && al16(x, x_stride, x_dt) && al16(mul, mul_stride, HICOM_DT_F32) && al16(add, add_stride, HICOM_DT_F32) &&
        al16(gamma, 0, gb_dt) && al16(beta, 0, gb_dt) && al16(src, src_stride, src_dt) && al16(out, out_stride, out_dt)) {
        hipLaunchKernelGGL(row_ln8_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
        return hicom_host::check_launch("row_ln");
    }
    if (E <= 768) hipLaunchKernelGGL(row_ln_kernel<12>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (E <= 1152) hipLaunchKernelGGL(row_ln_kernel<18>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(row_ln_kernel<kMaxPerLane>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return hicom_host::check_launch("row_ln");
}

extern "C" int hicom_small_mha_fwd(const float* q, const float* k, const float* v, int32_t M, int32_t L,
                                   int32_t nh, int32_t hd, float* out, void* stream) {
    return hicom_small_mha_scaled_fwd(q, k, v, M, L, nh, hd, hd > 0 ? 1.0f / sqrtf((float)hd) : 0.f, out, stream);
}

extern "C" int hicom_small_mha_scaled_fwd(const float* q, const float* k, const float* v, int32_t M, int32_t L,
                                          int32_t nh, int32_t hd, float scale, float* out, void* stream) {
    HICOM_REQUIRE(q && k && v && out, HICOM_EINVAL, "small_mha: NULL pointer");
    HICOM_REQUIRE(M > 0 && L > 0 && L <= 64 && nh > 0 && hd > 0, HICOM_EUNSUP, "small_mha: L=%d (<= 64 keys supported)", L);
    if ((hd == 128 || hd == 96 || hd == 64) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0)) {
        const size_t smem = (size_t)(64 * (hd + 4) + 64 * hd) * 4;
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(small_mha_lds_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (64 * 132 + 64 * 128) * 4);
            attr_set = true;
        }
        const dim3 grid((unsigned)nh, (unsigned)((M + kMhaRows - 1) / kMhaRows));
        const float sc = scale;
        hipStream_t st = (hipStream_t)stream;
        if (hd == 128) hipLaunchKernelGGL(small_mha_lds_kernel<128>, grid, dim3(256), smem, st, q, k, v, M, L, nh, sc, out);
        else if (hd == 96) hipLaunchKernelGGL(small_mha_lds_kernel<96>, grid, dim3(256), smem, st, q, k, v, M, L, nh, sc, out);
        else hipLaunchKernelGGL(small_mha_lds_kernel<64>, grid, dim3(256), smem, st, q, k, v, M, L, nh, sc, out);
        return hicom_host::check_launch("small_mha");
    }
    const long waves = (long)M * nh;
    hipLaunchKernelGGL(small_mha_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, k, v, M, L,
                       nh, hd, scale, out);
    return hicom_host::check_launch("small_mha");
}
