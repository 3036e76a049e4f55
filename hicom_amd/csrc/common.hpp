// Shared device helpers for the gfx950 (CDNA4, wave64) HICom kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/hicom_hip.h"

namespace hicom {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one 16x16x32 A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;   // 4 bf16 = one 16x16x16 A/B fragment
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 f32 accumulator fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(3))) unsigned int u32x3;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t packed) { return __uint_as_float(packed & 0xFFFF0000u); }

// round-to-nearest-even f32 -> bf16 bits (plain cast: hipcc emits v_cvt_pk_bf16_f32, NaN-safe)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}

// x ~= hi + lo, both bf16; |x - hi - lo| <= 2^-17 |x|
__device__ __forceinline__ void split_bf16(float x, uint16_t& hi, uint16_t& lo) {
    hi = f32_to_bf16(x);
    lo = f32_to_bf16(x - bf16_to_f32(hi));
}

// Reductions over the 16 lanes of a DPP row (lanes 16k..16k+15) by row rotations: four VALU ops, every
// lane ends up with the result.  (__shfl_xor compiles to ds_bpermute_b32: an LDS round trip per step.)
template <int ROR>
__device__ __forceinline__ float dpp_row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + ROR, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_row_ror<8>(v));
    v = fmaxf(v, dpp_row_ror<4>(v));
    v = fmaxf(v, dpp_row_ror<2>(v));
    return fmaxf(v, dpp_row_ror<1>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    return v + dpp_row_ror<1>(v);
}

// All-reduce over the four lanes l, l^16, l^32, l^48 (one lane of each 16-lane DPP row) with the gfx950
// row swaps: v_permlane16_swap pairs rows {0,1} and {2,3}, v_permlane32_swap the two halves.  Pure VALU.
__device__ __forceinline__ float xrow4_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xrow4_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// sum over the 32 lanes of a half wave (rows {0,1} or {2,3}): DPP rows, then one v_permlane16_swap
__device__ __forceinline__ float half32_sum(float v) {
    v = row16_sum(v);
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float wave_max_fast(float v) { return xrow4_max(row16_max(v)); }

// whole-wave sum without LDS: DPP row rotations (16-lane rows), then the two gfx950 row swaps.  __shfl_xor compiles to
// ds_bpermute_b32, an LDS round trip per step: six dependent ones per reduction.
__device__ __forceinline__ float wave_sum_fast(float v) { return xrow4_sum(row16_sum(v)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exp(x) for x <= 0 as v_exp_f32(x * log2 e): 2 instructions instead of ~20.  Relative error
// <= 2^-22 + |x| * 2^-24 (argument rounding), i.e. < 2e-6 for the softmax range |x| <= 30 -- three
// orders below the 1e-3 parity budget.  x = -1e30 (masked / first tile) gives exactly 0.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// erf-GELU (nn.GELU() default, reference projector.py:309).  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, one
// exp2 + one rcp): libm's erff is ~60 VALU instructions, which in a GEMM epilogue with one wave per SIMD was a
// quarter of the kernel.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float tail = poly * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);   // 1 - erf(z), z >= 0
    // 1 + erf(x/sqrt2) = 2 - tail (x >= 0) or tail (x < 0)
    return 0.5f * x * (x >= 0.f ? 2.0f - tail : tail);
}

// start of window i along an axis (projector.py:501-522 restated in closed form)
__device__ __host__ __forceinline__ int axis_start(const hicom_axis& a, int i) {
    return i < a.nfull ? i * a.k : a.nfull * a.k + (i - a.nfull) * (a.k - 1) - 1;
}

// raw workgroup barrier that orders LDS traffic only (leaves global loads / LDS-DMA in flight)
// LDS float add without return, issued as inline asm: through atomicAdd hipcc orders it behind ALL
// outstanding vector-memory traffic (s_waitcnt vmcnt(0)), which drains LDS-DMA prefetches in flight.
// Completion is covered by the next lds_barrier() (lgkmcnt(0)); same-wave LDS ops stay in issue order.
__device__ __forceinline__ void lds_add_f32(float* lds_ptr, float v) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lds_ptr);
    asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace hicom

// host-side error plumbing shared by the C-ABI translation units
namespace hicom_host {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// An event to be recorded right behind the NEXT kernel launch of this thread, folded into that launch
// (hipExtLaunchKernelGGL's stop event) instead of a separate hipEventRecord: every host call costs 2-4 us and the
// executor's call sequence is what bounds the step on slow hosts.  Consumed by HICOM_LAUNCH.
void set_stop_event(void* ev);
void* take_stop_event();
}  // namespace hicom_host

#define HICOM_LAUNCH(kernel, grid, block, smem, stream, ...)                                                      \
    do {                                                                                                         \
        void* stop_ev_ = hicom_host::take_stop_event();                                                          \
        if (stop_ev_) hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, nullptr, (hipEvent_t)stop_ev_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                 \
    } while (0)

#define HICOM_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            hicom_host::set_error(__VA_ARGS__);   \
            return (code);                        \
        }                                         \
    } while (0)
