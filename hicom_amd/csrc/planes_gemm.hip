// Readout MLP GEMM on bf16 "planes": y = act((a_hi + a_lo) . w^T + b).
//
// fp32 intermediates that feed matrix cores travel between kernels as TWO bf16 matrices (hi = RNE
// bf16 of the value, lo = bf16 of the remainder; hi + lo carries 16 mantissa bits).  The producer
// (fused ring kernel / previous GEMM) writes the planes, so this GEMM has no conversion work at
// all: every operand tile goes HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) through a
// five-stage ring, counted s_waitcnt vmcnt + raw s_barrier (a __syncthreads would drain the DMA).
//
// Replaces nn.Linear / nn.GELU / nn.Linear of build_mlp on the window tokens
// (reference projector.py:307-312, :559).
//
// The problem is small (1296 x 896 outputs) and on this part it is bound by what one CU can pull
// through its 64 B/clk vector-memory path and its 128 B/clk LDS, not by MFMA or HBM: so ONE
// 256-thread workgroup per CU owns a 48 x 128 tile (27 x 7 = 189 tiles: every tile is resident at
// once, no second round and no CU with two tiles), and each wave owns a 48 x 64 sub-tile of ONE
// 32-wide half of every BK = 64 stage (waves = 2 column halves x 2 k halves) -- 10 fragment reads
// feed 24 MFMAs, where a 32 x 32 wave tile needed 12 for 16.  The two k halves meet once, in the
// epilogue, through the (then idle) ring memory.
// LDS stage = A_hi | A_lo | W images of [rows][128 B]; the 16-byte chunk index is XORed with
// (row >> 1) & 7 on the DMA *source* address and on the fragment read (the DMA destination is
// lane-linear): the 16 rows of a ds_read_b128 quarter-wave hit 16 different bank groups.
// The product is computed transposed (W fragment as the A operand): a lane then holds 4 consecutive
// output columns of one row, i.e. 8-byte bf16 / 16-byte fp32 stores.  Fragment reads of stage s+1
// are issued before the MFMAs of stage s (two register sets), XCD-aware tile order as in
// readout_gemm.hip.
#include <type_traits>

#include "common.hpp"

namespace hicom {

struct PlanesGemmParams {
    const uint16_t* a_hi;
    const uint16_t* a_lo;
    const uint16_t* w;
    const void* b;
    int b_f32;
    int M, N, K, act;
    // output A: bf16 planes [M][N] (hidden activations for the next GEMM)
    uint16_t* o_hi;
    uint16_t* o_lo;
    // output B: packed rows of the final tensor (dtype y_f32 ? f32 : bf16)
    void* y;
    int y_f32;
    long ldy, row0;
    int nl_group;
    int vec;          // 4-column groups may be stored as one vector (alignment checked on the host)
    int bvec;         // ... and the bias may be loaded as one vector per 4 columns
};

constexpr int kPTM = 48, kPTN = 128;
constexpr int kPImgA = kPTM * 128;                  // bytes of one activation image (48 rows x 64 bf16)
constexpr int kPStage = 2 * kPImgA + kPTN * 128;    // 28 KB: A_hi | A_lo | W
constexpr int kPRing = 5;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// HAS_LO = false: the activation is exactly bf16 (raw visual tokens feeding the k/v adaptor MLPs): no
// lo plane is fetched and half the MFMAs are issued.
template <bool HAS_LO>
__global__ __launch_bounds__(256, 2) void planes_gemm_kernel(PlanesGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [kPRing][kPStage]
    constexpr int PW = HAS_LO ? 7 : 6;                           // DMA pieces per wave per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 1, ch = wave & 1;
    const int r16 = lane & 15, kg = lane >> 4;
    const int nbx = (p.N + kPTN - 1) / kPTN, nby = (p.M + kPTM - 1) / kPTM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int by = xcd + 8 * (slot / nbx), bx = slot - (slot / nbx) * nbx;
    if (by >= nby) return;
    const int m0 = by * kPTM, n0 = bx * kPTN;
    const int ns = p.K >> 6;

    // DMA assignment: a stage is 28 one-KiB pieces (8 rows x 128 B each): 6 A_hi, 6 A_lo, 16 W; wave w
    // issues pieces w, w+4, ...  Without a lo plane 22 pieces remain: 24 slots, two harmless repeats.
    const int prow = lane >> 3, cpos = lane & 7;
    const uint16_t* src[PW];
    int dst_off[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        int pi = wave + 4 * i;
        if (!HAS_LO) {
            if (pi >= 22) pi -= 22;
            if (pi >= 6) pi += 6;
        }
        if (pi < 12) {
            const int row = 8 * (pi < 6 ? pi : pi - 6) + prow;
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            src[i] = (pi < 6 ? p.a_hi : p.a_lo) + (long)m * p.K + 8 * (cpos ^ ((row >> 1) & 7));
        } else {
            const int row = 8 * (pi - 12) + prow;
            int n = n0 + row;
            n = n < p.N ? n : p.N - 1;
            src[i] = p.w + (long)n * p.K + 8 * (cpos ^ ((row >> 1) & 7));
        }
        dst_off[i] = pi * 1024;
    }
    auto issue = [&](int s, int ring_slot) {
        char* base = lds + ring_slot * kPStage;
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + 64 * s),
                                             (__attribute__((address_space(3))) void*)(base + dst_off[i]), 16, 0, 0);
    };
    // wait until at most k of this wave's stages are still in flight
    auto wait_stages = [&](int k) {
        if (k >= 2) wait_vm<2 * PW>();
        else if (k == 1) wait_vm<PW>();
        else wait_vm<0>();
    };

    // fragment addresses (this wave's k half).  acc[jl] is column block jn = jl ^ (2 * kh): the blocks a
    // wave keeps in the epilogue are always jl = 0, 1.
    const int frag_a = r16 * 128 + 16 * ((4 * kh + kg) ^ ((r16 >> 1) & 7));
    const int frag_w0 = 2 * kPImgA + (64 * ch + 32 * kh) * 128 + frag_a;          // jl = 0, 1
    const int frag_w1 = 2 * kPImgA + (64 * ch + 32 * (1 - kh)) * 128 + frag_a;    // jl = 2, 3
    struct Frags {
        bf16x8 w[4], ah[3], al[3];
    };
    // Fragment reads are inline asm: hipcc's own waitcnt bookkeeping falls back to lgkmcnt(0) ahead of the first
    // MFMA of a step, i.e. it would wait for the reads that were just issued for the NEXT step.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(lds);
#define HICOM_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto read = [&](int ring_slot, Frags& f) {
        const unsigned st = lds0 + ring_slot * kPStage;
        const unsigned aa = st + frag_a, w0 = st + frag_w0, w1 = st + frag_w1;
        HICOM_LDS_RD(f.w[0], w0, 0);
        HICOM_LDS_RD(f.w[1], w0, 2048);
        HICOM_LDS_RD(f.ah[0], aa, 0);
        HICOM_LDS_RD(f.ah[1], aa, 2048);
        HICOM_LDS_RD(f.ah[2], aa, 4096);
        if (HAS_LO) {
            HICOM_LDS_RD(f.al[0], aa, kPImgA);
            HICOM_LDS_RD(f.al[1], aa, kPImgA + 2048);
            HICOM_LDS_RD(f.al[2], aa, kPImgA + 4096);
        }
        HICOM_LDS_RD(f.w[2], w1, 0);
        HICOM_LDS_RD(f.w[3], w1, 2048);
    };
#undef HICOM_LDS_RD
    auto land = [&](Frags& f) {
        if (HAS_LO)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]), "+v"(f.ah[0]), "+v"(f.ah[1]), "+v"(f.ah[2]),
                           "+v"(f.al[0]), "+v"(f.al[1]), "+v"(f.al[2])::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]), "+v"(f.ah[0]), "+v"(f.ah[1]), "+v"(f.ah[2])::"memory");
    };

    f32x4 acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const Frags& f) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[j], f.ah[i], acc[j][i], 0, 0, 0);
        if (HAS_LO) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[j], f.al[i], acc[j][i], 0, 0, 0);
        }
    };

    // prologue: stages 0 .. kPRing-2 in flight, stage 0 landed, its fragments on the way
    const int npro = ns < kPRing - 1 ? ns : kPRing - 1;
    for (int s = 0; s < npro; ++s) issue(s, s);
    // the bias of this wave's epilogue columns, fetched now (raw, one vector load per column block): a dependent load
    // in the epilogue is ~1 us of a 13-us kernel
    uint2 braw[2] = {make_uint2(0, 0), make_uint2(0, 0)};
    float4 brawf[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    bool bpre[2];
#pragma unroll
    for (int jl = 0; jl < 2; ++jl) {
        const int nb = n0 + 64 * ch + 16 * (2 * kh + jl) + 4 * kg;
        bpre[jl] = p.b && p.bvec && nb + 3 < p.N;
        if (bpre[jl]) {
            if (p.b_f32) brawf[jl] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.b) + nb);
            else braw[jl] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.b) + nb);
        }
    }
    if (npro == kPRing - 1) wait_vm<(kPRing - 2) * PW>();
    else wait_stages(npro - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frags f0, f1;
    read(0, f0);
    land(f0);
    int slot_next = 1;                 // ring slot of stage s+1
    int slot_issue = kPRing - 1;       // ring slot of stage s+kPRing-1 (= the slot of stage s-1)
    // one step (s + 1 < ns): stage s+1 has landed for every wave and every wave is past its reads of stage s-1;
    // the fragment reads of stage s+1 then fly under the MFMAs of stage s
    auto step = [&](auto steady, int s, const Frags& cur, Frags& nxt) {
        if constexpr (decltype(steady)::value) {
            // s + kPRing - 1 < ns: branch-free, so that hipcc keeps its counted lgkmcnt waits
            wait_vm<(kPRing - 3) * PW>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(s + kPRing - 1, slot_issue);
        } else {
            const int ahead = (ns - 1 < s + kPRing - 2 ? ns - 1 : s + kPRing - 2) - (s + 1);
            wait_stages(ahead);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + kPRing - 1 < ns) issue(s + kPRing - 1, slot_issue);
        }
        read(slot_next, nxt);
        __builtin_amdgcn_sched_barrier(0);      // the reads of stage s+1 fly under the MFMAs of stage s
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        // ... and have landed long before the 24 MFMAs are through: closing them here keeps asynchronous register
        // writes (which the compiler knows nothing about) from crossing a control-flow edge
        land(nxt);
        slot_next = slot_next + 1 == kPRing ? 0 : slot_next + 1;
        slot_issue = slot_issue + 1 == kPRing ? 0 : slot_issue + 1;
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    int s = 0;
    for (; s + kPRing < ns; s += 2) {
        step(Yes{}, s, f0, f1);
        step(Yes{}, s + 1, f1, f0);
    }
    for (; s + 2 < ns; s += 2) {
        step(No{}, s, f0, f1);
        step(No{}, s + 1, f1, f0);
    }
    if (ns - s == 2) {
        step(No{}, s, f0, f1);
        compute(f1);
    } else {
        compute(f0);
    }

    // the k halves meet: a wave hands column blocks jl = 2, 3 to its partner and keeps jl = 0, 1
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // MFMA results -> LDS/VALU reads: do not rely on hipcc's padding
    __builtin_amdgcn_s_barrier();                   // every wave is done reading the ring
    asm volatile("" ::: "memory");
    float4* xb = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const f32x4 v = acc[2 + t / 3][t % 3];
        xb[(wave * 6 + t) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    lds_barrier();
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const float4 v = xb[((wave ^ 2) * 6 + t) * 64 + lane];
        acc[t / 3][t % 3][0] += v.x;
        acc[t / 3][t % 3][1] += v.y;
        acc[t / 3][t % 3][2] += v.z;
        acc[t / 3][t % 3][3] += v.w;
    }

    // epilogue.  Transposed product: lane holds columns n .. n+3 (4 * kg + q) of row m = r16.
#pragma unroll
    for (int jl = 0; jl < 2; ++jl) {
        const int n = n0 + 64 * ch + 16 * (2 * kh + jl) + 4 * kg;
        if (n >= p.N) continue;
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (bpre[jl]) {
            if (p.b_f32) {
                bias[0] = brawf[jl].x; bias[1] = brawf[jl].y; bias[2] = brawf[jl].z; bias[3] = brawf[jl].w;
            } else {
                bias[0] = bf16lo_to_f32(braw[jl].x); bias[1] = bf16hi_to_f32(braw[jl].x);
                bias[2] = bf16lo_to_f32(braw[jl].y); bias[3] = bf16hi_to_f32(braw[jl].y);
            }
        } else if (p.b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nn = n + q < p.N ? n + q : p.N - 1;
                bias[q] = p.b_f32 ? reinterpret_cast<const float*>(p.b)[nn] : bf16_to_f32(reinterpret_cast<const uint16_t*>(p.b)[nn]);
            }
        }
        const bool vec = p.vec && n + 3 < p.N;
#pragma unroll
        for (int im = 0; im < 3; ++im) {
            const int m = m0 + 16 * im + r16;
            if (m >= p.M) continue;
            float v[4];
            uint16_t h[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = acc[jl][im][q] + bias[q];
                if (p.act == HICOM_ACT_GELU) v[q] = gelu_erf(v[q]);
            }
            if (p.o_hi) {
#pragma unroll
                for (int q = 0; q < 4; ++q) split_bf16(v[q], h[q], l[q]);
                uint16_t* oh = p.o_hi + (long)m * p.N + n;
                uint16_t* ol = p.o_lo + (long)m * p.N + n;
                if (vec) {
                    *reinterpret_cast<uint2*>(oh) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
                    *reinterpret_cast<uint2*>(ol) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
                } else {
                    for (int q = 0; q < 4 && n + q < p.N; ++q) {
                        oh[q] = h[q];
                        ol[q] = l[q];
                    }
                }
            }
            if (p.y) {
                const long orow = p.row0 + m + (p.nl_group > 0 ? m / p.nl_group : 0);
                if (p.y_f32) {
                    float* yo = reinterpret_cast<float*>(p.y) + orow * p.ldy + n;
                    if (vec) *reinterpret_cast<float4*>(yo) = make_float4(v[0], v[1], v[2], v[3]);
                    else
                        for (int q = 0; q < 4 && n + q < p.N; ++q) yo[q] = v[q];
                } else {
                    uint16_t* yo = reinterpret_cast<uint16_t*>(p.y) + orow * p.ldy + n;
                    if (vec) {
                        const uint32_t u0 = f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                        const uint32_t u1 = f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(yo) = make_uint2(u0, u1);
                    } else {
                        for (int q = 0; q < 4 && n + q < p.N; ++q) yo[q] = f32_to_bf16(v[q]);
                    }
                }
            }
        }
    }
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_planes_gemm_fwd(const void* a_hi, const void* a_lo, const void* w, const void* b, int32_t b_dt,
                                     int32_t M, int32_t N, int32_t K, int32_t act,
                                     void* out_hi, void* out_lo,
                                     void* y, int32_t y_dt, int64_t ldy, int64_t row0, int32_t nl_group, void* stream) {
    HICOM_REQUIRE(a_hi && w, HICOM_EINVAL, "planes_gemm: NULL pointer");
    HICOM_REQUIRE((out_hi && out_lo) || y, HICOM_EINVAL, "planes_gemm: no output");
    HICOM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, HICOM_EINVAL, "planes_gemm: bad shape M=%d N=%d K=%d (K %% 64)", M, N, K);
    HICOM_REQUIRE(!y || (ldy >= N && row0 >= 0 && nl_group >= 0), HICOM_EINVAL, "planes_gemm: bad output layout");
    HICOM_REQUIRE(((uintptr_t)a_hi % 16 == 0) && ((uintptr_t)a_lo % 16 == 0) && ((uintptr_t)w % 16 == 0) && M < (1 << 30), HICOM_EINVAL,
                  "planes_gemm: alignment");
    const bool vec = N % 4 == 0 && (!out_hi || (((uintptr_t)out_hi | (uintptr_t)out_lo) % 8 == 0)) &&
                     (!y || (ldy % 4 == 0 && (uintptr_t)y % 16 == 0));
    PlanesGemmParams p{(const uint16_t*)a_hi, (const uint16_t*)a_lo, (const uint16_t*)w, b, b_dt == HICOM_DT_F32,
                       M, N, K, act, (uint16_t*)out_hi, (uint16_t*)out_lo, y, y_dt == HICOM_DT_F32, (long)ldy, (long)row0,
                       nl_group, vec ? 1 : 0, (b && (uintptr_t)b % 16 == 0) ? 1 : 0};
    const int nbx = (N + kPTN - 1) / kPTN, nby = (M + kPTM - 1) / kPTM;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(planes_gemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kPRing * kPStage);
        hipFuncSetAttribute(reinterpret_cast<const void*>(planes_gemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kPRing * kPStage);
        attr_set = true;
    }
    const dim3 grid((unsigned)(8 * nbx * ((nby + 7) / 8)));
    if (a_lo) HICOM_LAUNCH(planes_gemm_kernel<true>, grid, dim3(256), kPRing * kPStage, (hipStream_t)stream, p);
    else HICOM_LAUNCH(planes_gemm_kernel<false>, grid, dim3(256), kPRing * kPStage, (hipStream_t)stream, p);
    return hicom_host::check_launch("planes_gemm");
}
