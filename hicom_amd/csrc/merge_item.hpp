// merge + v_proj of the ring kernel's partial global states, one (head, 64-channel slab) ITEM at a time -- shared by the
// standalone launch (merge.hip: hicom_merge_vproj_fixed_fwd) and by the merge ROLE inside readout GEMM 1's launch
// (readout16.hip, round 5: the merge is independent of the local readout, so it rides on the CUs the tile grid leaves idle
// and its launch disappears from the step).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace hicom {

// fp16 halves of a dword as f32.  NOT `__builtin_bit_cast(half2, vec[q])`: hipcc (ROCm 7.2) reads ELEMENT 0 for every q when a
// bit_cast is applied to an element of an ext_vector (the same toolchain bug as DESIGN.md §3.2's fdot2 note: tools/marg_role_debug.py
// showed every dword of a 16-byte load decoding as its first) -- shift the scalar instead.
__device__ __forceinline__ float mv_f16lo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xFFFFu)); }
__device__ __forceinline__ float mv_f16hi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }

__device__ __forceinline__ float mv_block_max(float v, float* red) {
    v = wave_max_fast(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float mv_block_sum(float v, float* red) {
    v = wave_sum_fast(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr float kMvFixScale = 68719476736.f;          // 2^36: |o| < 2^26 representable, 1.5e-11 resolution

struct MergeVprojFixParams {
    const float* part_m;
    const float* part_l;
    const void* part_acc;   // fp32 un-normalised accumulators, or (F16) fp16 normalised contexts
    int nparts, rows_pad, E, hd;
    const uint16_t* wv;     // bf16 [E, E]
    long long* o_fix;       // [E] fixed-point accumulators (zero on entry)
    float* out_ml;          // [R][2] or NULL
    float* out_ctx;         // [R][E] normalised (or, ctx_unnorm: the un-normalised accumulator relative to M), or NULL
    // layout of the partials (elements): m / l of (part i, row h) at part_m[i * ml_part + h * ml_row]; accumulator row at
    // part_acc[i * acc_part + h * acc_row].  Defaults (0): ml_part = rows_pad, ml_row = 1, acc_part = rows_pad * E, acc_row = E.
    long ml_part, ml_row, acc_part, acc_row;
    int ctx_unnorm;         // out_ctx receives sum_i w_i ACC_i (the shard STATE of the frame-sharded path) instead of the normalised context
    // Round 6: the value-side pos-emb rides HERE instead of behind the token stream of the ring kernel.  By linearity of v_proj,
    //   W_v,h (ctx_h + sum_s mg_h[s] pe[s]) = W_v,h ctx_h + sum_s mg_h[s] VPE[h hd + j][s],   VPE = W_v . PE^T (weight-only, like kpe),
    // with mg_h[s] the t / y / x marginals of head h's merged softmax weights over the absolute slots [T | H | W | padding].  The ring
    // leaves them per partial, normalised (marginal / l, fp16: RingParams::part_marg); item (h, slab k) merges the 8 slots
    // [8k, 8k + 8) with the weights it has anyway and adds their VPE products to its rows' dot products IN FRONT of the fixed-point
    // conversion -- no extra atomics, two 16-byte loads per thread.  marg_slots must be 8 * (E / kMvSlab).  NULL: pos-emb already in
    // the partial contexts (or none).
    const _Float16* part_marg;   // [nparts][E / hd][marg_slots]
    const _Float16* vpe16;       // [E][marg_slots]
    int marg_slots;
};

template <int kMvSlab>
constexpr int mv_item_lds_bytes() { return (256 + 4 + (256 / (kMvSlab / 4)) * kMvSlab + kMvSlab + 4 * 8 + 8 + 256 * 4) * 4; }

// One item: head h, channels [slab * kMvSlab, + kMvSlab).  256 threads; `lds` >= mv_item_lds_bytes<kMvSlab>() bytes, 16-byte aligned.
// Split into a LOAD half (every global read of the item, into registers) and a COMPUTE half, so that a workgroup that owns several
// items (the merge role inside readout GEMM 1's launch) requests all of them before it touches the first: its items then cost ONE
// memory round trip instead of one each (beside ~200 streaming tile workgroups a round trip is 3-5 us).
template <int kMvSlab, bool F16>
struct MvItemRegs {
    static constexpr int NC4 = kMvSlab / 4, NG = 256 / NC4, NU = (256 + NG - 1) / NG, WQ = kMvSlab / 16;
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    u32x4 wreg[WQ];                               // v_proj weights of this thread's (row j, half of the slab)
    typename std::conditional<F16, half4_t, float4>::type v[NU];   // raw partial rows
    float pm, pl;
    u32x4 mg, vp;                                 // 8 fp16 marginals of partial `tid`; 8 fp16 VPE entries of row j (threads with half == 0)
};

template <int kMvSlab, bool F16>      // channels per item (32 or 64); partials as normalised fp16 contexts
__device__ __forceinline__ void merge_vproj_fixed_load(const MergeVprojFixParams& p, int slab, int h, MvItemRegs<kMvSlab, F16>& r) {
    using R = MvItemRegs<kMvSlab, F16>;
    const int tid = threadIdx.x;
    // v_proj weights of this (head, slab) first: independent of everything else here (wv NULL: merge only, no v_proj)
    const int j = tid >> 1, half = tid & 1;
#pragma unroll
    for (int q = 0; q < R::WQ; ++q)
        r.wreg[q] = (p.wv && j < p.hd) ? *reinterpret_cast<const u32x4*>(p.wv + (long)(h * p.hd + j) * p.E + slab * kMvSlab + (kMvSlab / 2) * half + 8 * q) : u32x4{0, 0, 0, 0};
    // raw partial rows: thread = (float4 column c4 of the slab, partial group pg of NG); all of a thread's <= NU loads in flight
    const int c4 = tid % R::NC4, pg = tid / R::NC4;
    const long eoff = (long)h * (p.acc_row ? p.acc_row : p.E) + slab * kMvSlab + 4 * c4;
    const long pstride = p.acc_part ? p.acc_part : (long)p.rows_pad * p.E;
    const long mlp = p.ml_part ? p.ml_part : p.rows_pad, mlr = p.ml_row ? p.ml_row : 1;
#pragma unroll
    for (int u = 0; u < R::NU; ++u) {
        const int i = pg + R::NG * u;
        if constexpr (F16) {
            r.v[u] = typename R::half4_t{0, 0, 0, 0};
            if (i < p.nparts) r.v[u] = *reinterpret_cast<const typename R::half4_t*>(reinterpret_cast<const _Float16*>(p.part_acc) + eoff + (long)i * pstride);
        } else {
            const float* src = reinterpret_cast<const float*>(p.part_acc) + eoff + (long)i * pstride;
            if (i >= p.nparts) {
                r.v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (p.acc_row) {                        // shard states: the accumulators start 2 * rows floats into a set (8-byte aligned)
                const float2 lo = *reinterpret_cast<const float2*>(src), hi = *reinterpret_cast<const float2*>(src + 2);
                r.v[u] = make_float4(lo.x, lo.y, hi.x, hi.y);
            } else {
                r.v[u] = *reinterpret_cast<const float4*>(src);
            }
        }
    }
    r.pm = tid < p.nparts ? p.part_m[(long)tid * mlp + h * mlr] : -1.0e30f;       // nparts <= 256 (host-checked)
    r.pl = tid < p.nparts ? p.part_l[(long)tid * mlp + h * mlr] : 0.f;
    // (straight-line, unconditional loads with clamped indices: a load inside a branch makes hipcc wait for EVERYTHING in flight at the
    // join, and a role workgroup's three items then cost three round trips instead of one -- GEMM 1's launch 15.1 against 12.1 us.
    // Without marginals both pointers fall back to the v_proj weights: valid memory, the values are dropped by the compute half.)
    {
        const int nrow = p.E / p.hd;
        const int ti = tid < p.nparts ? tid : p.nparts - 1, jj = j < p.hd ? j : p.hd - 1;
        const char* mgp = p.part_marg ? reinterpret_cast<const char*>(p.part_marg + ((long)ti * nrow + h) * p.marg_slots + 8 * slab)
                                      : reinterpret_cast<const char*>(p.wv ? (const void*)p.wv : (const void*)p.part_m);
        const char* vpp = p.part_marg ? reinterpret_cast<const char*>(p.vpe16 + (long)(h * p.hd + jj) * p.marg_slots + 8 * slab)
                                      : reinterpret_cast<const char*>(p.wv ? (const void*)p.wv : (const void*)p.part_m);
        r.mg = *reinterpret_cast<const u32x4*>(mgp);
        r.vp = *reinterpret_cast<const u32x4*>(vpp);
    }
}

template <int kMvSlab, bool F16>
__device__ __forceinline__ void merge_vproj_fixed_compute(const MergeVprojFixParams& p, int slab, int h, const MvItemRegs<kMvSlab, F16>& r, char* lds) {
    using R = MvItemRegs<kMvSlab, F16>;
    float* cpart = reinterpret_cast<float*>(lds);            // [NG][kMvSlab]
    float* cx = cpart + R::NG * kMvSlab;                      // [kMvSlab]
    float* wp = cx + kMvSlab;                                 // [256]
    float* red = wp + 256;                                    // [4]
    float* mgw = red + 4;                                     // [4 waves][8] partial sums of this item's 8 marginal slots
    float* mgn = mgw + 32;                                    // [8] merged, normalised marginals
    u32x4* mgl = reinterpret_cast<u32x4*>(mgn + 8);           // [256] the partials' 8 fp16 marginals, one 16-byte entry per partial
    const int tid = threadIdx.x;
    const int j = tid >> 1, half = tid & 1;
    const int c4 = tid % R::NC4, pg = tid / R::NC4;
    const float pm = r.pm, pl = r.pl;
    const float M = mv_block_max(pm, red);
    const float w = tid < p.nparts ? expf(pm - M) : 0.f;
    wp[tid] = F16 ? w * pl : w;                          // (normalised contexts are weighed with l e^(m - M))
    if (p.part_marg) mgl[tid] = r.mg;                    // (visible behind the barriers below, like wp[])
    const float L = mv_block_sum(w * pl, red);           // (barriers inside: wp[] is visible afterwards)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < R::NU; ++u) {
        const int i = pg + R::NG * u;
        const float wu = i < 256 ? wp[i] : 0.f;
        float4 vv;
        if constexpr (F16) vv = make_float4((float)r.v[u][0], (float)r.v[u][1], (float)r.v[u][2], (float)r.v[u][3]);
        else vv = r.v[u];
        a.x = fmaf(wu, vv.x, a.x); a.y = fmaf(wu, vv.y, a.y); a.z = fmaf(wu, vv.z, a.z); a.w = fmaf(wu, vv.w, a.w);
    }
    *reinterpret_cast<float4*>(&cpart[pg * kMvSlab + 4 * c4]) = a;
    if (p.part_marg) {
        // marginals of this item's 8 slots: sum_i (l_i e^(m_i - M)) (mg_i / l_i).  Transposed through LDS: thread (slot q = tid % 8, group
        // g = tid / 8) sums the partials g, g + 32, ... (8 of them), a wave then holds 8 groups x 8 slots and three lane-exchange steps
        // finish its share -- eight whole-wave reductions per item (one per slot, ~100 dependent VALU steps, three items per role workgroup)
        // lengthened readout GEMM 1's launch by 1.1 us
        const int q = tid & 7, g = tid >> 3;
        const unsigned short* mh = reinterpret_cast<const unsigned short*>(mgl);
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = g + 32 * u;                                       // i < 256; partials >= nparts carry weight 0 and a clamped (finite) entry
            s = fmaf(wp[i], (float)__builtin_bit_cast(_Float16, mh[8 * i + q]), s);
        }
        // lanes of a wave: q = lane % 8, group = lane / 8 (8 groups): sum over the groups
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if ((tid & 63) < 8) mgw[(tid >> 6) * 8 + q] = s;
    }
    __syncthreads();
    if (p.part_marg && tid < 8) mgn[tid] = ((mgw[tid] + mgw[8 + tid]) + (mgw[16 + tid] + mgw[24 + tid])) / L;
    if (tid < kMvSlab) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < R::NG; ++g) sum += cpart[g * kMvSlab + tid];
        const float val = sum / L;
        cx[tid] = val;
        if (p.out_ctx) p.out_ctx[(long)h * p.E + slab * kMvSlab + tid] = p.ctx_unnorm ? sum : val;
    }
    if (tid == 0 && slab == 0 && p.out_ml) {
        p.out_ml[2 * h] = M;
        p.out_ml[2 * h + 1] = L;
    }
    __syncthreads();
    if (!p.wv) return;                                  // (merge only: the shard state of the frame-sharded path)
    // partial v_proj: thread (j, half) dots its half of the slab with weight row h*hd + j
    float dot = 0.f;
    if (j < p.hd) {
#pragma unroll
        for (int q = 0; q < R::WQ; ++q) {
            const u32x4 g = r.wreg[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dot = fmaf(bf16lo_to_f32(g[i]), cx[(kMvSlab / 2) * half + 8 * q + 2 * i], dot);
                dot = fmaf(bf16hi_to_f32(g[i]), cx[(kMvSlab / 2) * half + 8 * q + 2 * i + 1], dot);
            }
        }
    }
    if (p.part_marg && half == 0 && j < p.hd) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned u = r.vp[q];
            dot = fmaf(mgn[2 * q], mv_f16lo(u), dot);
            dot = fmaf(mgn[2 * q + 1], mv_f16hi(u), dot);
        }
    }
    dot += __shfl_xor(dot, 1, 64);
    if (half == 0 && j < p.hd) {
        const long long q = (long long)rintf(dot * kMvFixScale);
        __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned long long*)(p.o_fix + h * p.hd + j), (unsigned long long)q,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // result unused: no-return atomic
    }
}

template <int kMvSlab, bool F16>
__device__ __forceinline__ void merge_vproj_fixed_item(const MergeVprojFixParams& p, int slab, int h, char* lds) {
    MvItemRegs<kMvSlab, F16> r;
    merge_vproj_fixed_load<kMvSlab, F16>(p, slab, h, r);
    merge_vproj_fixed_compute<kMvSlab, F16>(p, slab, h, r, lds);
}

}  // namespace hicom
