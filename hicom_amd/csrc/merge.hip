// Merge of the streaming partials of the global compressor, the value-side positional term,
// and the cross-shard (cross-GPU) combine.
//
//   M_r = max_p m_p,  L_r = sum_p e^(m_p - M_r) l_p
//   ACC[r,:] = sum_p e^(m_p - M_r) acc_p[r,:]  +  sum_n e^(s_n - M_r) pos(n)
// with pos(n) = PE_t(t) + PE_y(y) + PE_x(x) (reference projector.py:95-99): the sum over tokens
// collapses to the t / y / x marginals of the softmax weights, so x + pos is never formed.
#include "common.hpp"

namespace hicom {

__global__ __launch_bounds__(64) void merge_stats_kernel(const float* part_m, const float* part_l, int nparts,
                                                         int rows_pad, float* ml) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float mx = -1.0e30f;
    for (int p = lane; p < nparts; p += 64) mx = fmaxf(mx, part_m[(long)p * rows_pad + r]);
    mx = wave_max(mx);
    float l = 0.f;
    for (int p = lane; p < nparts; p += 64) l += expf(part_m[(long)p * rows_pad + r] - mx) * part_l[(long)p * rows_pad + r];
    l = wave_sum(l);
    if (lane == 0) {
        ml[2 * r] = mx;
        ml[2 * r + 1] = l;
    }
}

// per (row, frame): marginals of e^(s - M) over y and x, and the frame total.
// scratch[(r*T + t) * (H + W + 1) + {0..H-1 | H..H+W-1 | H+W}]
__global__ __launch_bounds__(256) void frame_marginals_kernel(const float* scores, long score_stride, const float* ml,
                                                              int T, int H, int W, float* scratch) {
    extern __shared__ float e[];   // [H*W] + [H]
    const int r = blockIdx.x, t = blockIdx.y, HW = H * W;
    const float M = ml[2 * r];
    const float* s = scores + (long)r * score_stride + (long)t * HW;
    for (int i = threadIdx.x; i < HW; i += 256) e[i] = expf(s[i] - M);
    __syncthreads();
    float* out = scratch + ((long)r * T + t) * (H + W + 1);
    float* fy = e + HW;
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
    }
}

struct MergeCtxParams {
    const float* part_m;
    const float* part_acc;
    int nparts, rows_pad, E;
    const float* ml;
    const float* scratch;   // frame marginals or NULL
    const float* pe;
    int T, H, W, t0i, y0i, x0i;
    float* out_acc;
};

__global__ __launch_bounds__(256) void merge_ctx_kernel(MergeCtxParams p) {
    extern __shared__ float wsm[];   // [nparts] weights, then [T + H + W] positional weights
    const int r = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    const float M = p.ml[2 * r];
    float* wp = wsm;
    float* wpos = wsm + p.nparts;
    for (int i = threadIdx.x; i < p.nparts; i += 256) wp[i] = expf(p.part_m[(long)i * p.rows_pad + r] - M);
    if (p.scratch) {
        const int S = p.H + p.W + 1;
        const float* sc = p.scratch + (long)r * p.T * S;
        for (int j = threadIdx.x; j < p.T + p.H + p.W; j += 256) {
            float a = 0.f;
            if (j < p.T) a = sc[(long)j * S + p.H + p.W];
            else for (int t = 0; t < p.T; ++t) a += sc[(long)t * S + (j - p.T)];
            wpos[j] = a;
        }
    }
    __syncthreads();
    if (c >= p.E) return;
    float a = 0.f;
    for (int i = 0; i < p.nparts; ++i) a = fmaf(wp[i], p.part_acc[((long)i * p.rows_pad + r) * p.E + c], a);
    if (p.scratch) {
        float b = 0.f;
        for (int t = 0; t < p.T; ++t) b = fmaf(wpos[t], p.pe[(long)(p.t0i + t) * p.E + c], b);
        for (int y = 0; y < p.H; ++y) b = fmaf(wpos[p.T + y], p.pe[(long)(p.y0i + y) * p.E + c], b);
        for (int x = 0; x < p.W; ++x) b = fmaf(wpos[p.T + p.H + x], p.pe[(long)(p.x0i + x) * p.E + c], b);
        a += b;
    }
    p.out_acc[(long)r * p.E + c] = a;
}

__global__ __launch_bounds__(256) void combine_kernel(const float* ml, const float* acc, int nsets, int rows, int E,
                                                      float* ctx) {
    const int r = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= E) return;
    float M = -1.0e30f;
    for (int k = 0; k < nsets; ++k) M = fmaxf(M, ml[((long)k * rows + r) * 2]);
    float L = 0.f, a = 0.f;
    for (int k = 0; k < nsets; ++k) {
        const float w = expf(ml[((long)k * rows + r) * 2] - M);
        L = fmaf(w, ml[((long)k * rows + r) * 2 + 1], L);
        a = fmaf(w, acc[((long)k * rows + r) * E + c], a);
    }
    ctx[(long)r * E + c] = a / L;
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_global_merge_fwd(const float* part_m, const float* part_l, const float* part_acc,
                                      int32_t nparts, int32_t rows, int32_t rows_pad, int32_t E,
                                      const float* scores, int64_t score_stride, int64_t N,
                                      int32_t H, int32_t W, const float* pe,
                                      int32_t t_index0, int32_t y_index0, int32_t x_index0,
                                      float* scratch, float* out_ml, float* out_acc, void* stream) {
    HICOM_REQUIRE(part_m && part_l && part_acc && out_ml && out_acc, HICOM_EINVAL, "global_merge: NULL pointer");
    HICOM_REQUIRE(nparts > 0 && rows > 0 && rows <= rows_pad && E > 0, HICOM_EINVAL, "global_merge: bad shape");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(merge_stats_kernel, dim3((unsigned)rows), dim3(64), 0, s, part_m, part_l, nparts, rows_pad, out_ml);
    int T = 0;
    if (pe) {
        HICOM_REQUIRE(scores && scratch && H > 0 && W > 0 && N > 0 && N % ((long)H * W) == 0, HICOM_EINVAL,
                      "global_merge: positional term needs scores, scratch and N %% (H*W) == 0");
        T = (int)(N / ((long)H * W));
        const size_t smem = ((size_t)H * W + H) * 4;
        HICOM_REQUIRE(smem <= 60000, HICOM_EUNSUP, "global_merge: grid %dx%d too large for the marginal kernel", H, W);
        hipLaunchKernelGGL(frame_marginals_kernel, dim3((unsigned)rows, (unsigned)T), dim3(256), smem, s, scores,
                           (long)score_stride, out_ml, T, H, W, scratch);
    }
    MergeCtxParams p{part_m, part_acc, nparts, rows_pad, E, out_ml, pe ? scratch : nullptr, pe,
                     T, H, W, t_index0, y_index0, x_index0, out_acc};
    const size_t smem2 = ((size_t)nparts + (pe ? (size_t)T + H + W : 0)) * 4;
    HICOM_REQUIRE(smem2 <= 60000, HICOM_EUNSUP, "global_merge: too many partials/frames for one pass");
    hipLaunchKernelGGL(merge_ctx_kernel, dim3((unsigned)rows, (unsigned)((E + 255) / 256)), dim3(256), smem2, s, p);
    return hicom_host::check_launch("global_merge");
}

extern "C" int hicom_global_combine_fwd(const float* ml, const float* acc, int32_t nsets, int32_t rows,
                                        int32_t E, float* ctx, void* stream) {
    HICOM_REQUIRE(ml && acc && ctx, HICOM_EINVAL, "global_combine: NULL pointer");
    HICOM_REQUIRE(nsets > 0 && rows > 0 && E > 0, HICOM_EINVAL, "global_combine: bad shape");
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)rows, (unsigned)((E + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, ml, acc, nsets, rows, E, ctx);
    return hicom_host::check_launch("global_combine");
}
