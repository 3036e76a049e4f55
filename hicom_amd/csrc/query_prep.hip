// Query prep of the release recipe in ONE launch (use_guide = direct: one injected query row, the guide itself):
//
//   qp      = W_q g + b_q                                              (reference projector.py:180)
//   qt_h    = scale * W_k,h^T qp_h        -> bf16 hi / lo planes        (:181, :193-197 folded)
//   pos_a_h = scale * kpe_h^T qp_h        (score-side pos-emb, kpe = W_k . PE^T cached by the caller)
//   rows R .. 15 of the ring kernel's A operand = the guide (local query, :352-368)
//   r0      = G0 (b_o + g) + g_b0         (the guide-dependent part of the first global readout layer, see below)
//
// Round 2 ran q_proj and the fold as two dependent launches (4.8 + 4.7 us for 2 x 2.65 MB of weights: both are
// launch-latency-sized).  The fold of head h needs exactly the hd = 128 q_proj outputs of head h, so the hand-off between
// the two stages is ~1 KB per consumer workgroup: here both stages run in one grid and the q_proj outputs travel as 8-byte
// {epoch tag, value} granules (cdna_hip_programming.md Guideline 16, form R2: the data is the flag; sc1 stores, a
// relaxed sc1 poll sweep, no fence, no flag).  The consumers request their W_k / kpe tiles BEFORE they start polling, so
// the cold-memory latencies of the two stages overlap instead of adding up.
//
// Epoch without a memset and without a host-side salt (a kernel argument would be frozen under hipGraph replay): an
// arrival counter in the state block (read_epoch below).  State words start at zero (the caller zeroes them once).
//
// Dispatch order: producer roles have the LOWEST block indices, so even a partially resident grid (another stream's work
// on the chip) cannot have consumers spinning for producers that have not been dispatched.  Every spin is bounded.
//
// r0: with C = G0 . W_o (weight-only, cached by the caller like kpe) the global tail
//   pre = W_o o + b_o + g ;  hid = GELU(G0 pre + g_b0)      (projector.py:226, :646, :307-312)
// becomes hid = GELU(C o + r0): one dependent stage fewer behind the streaming kernel; r0 depends on the guide only and
// is computed here, off the critical path.
#include "common.hpp"

namespace hicom {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct PrepParams {
    const uint16_t* g;        // [E] bf16: the injected global query (= the guide)
    const uint16_t* lq;       // [E] bf16: the shared local query (= the guide)
    const uint16_t* wq;       // [E, E] bf16
    const uint16_t* bq;       // [E] bf16 or NULL
    const uint16_t* wk;       // [E, E] bf16
    const float* kpe;         // [E, P] f32 or NULL
    int E, nh, hd, P;
    float scale;
    uint16_t* qhi;            // [16, E]
    uint16_t* qlo;
    float* pos_a;             // [16, pos_stride]
    int pos_stride, R;
    const uint16_t* gw0;      // [hidden, E] bf16 or NULL (no r0)
    const uint16_t* gb0;      // [hidden]
    const uint16_t* bo;       // [E]
    int hidden;
    float* r0;                // [hidden]
    unsigned long long* gran; // [E] granules {tag << 32 | f32 bits}
    unsigned* state;          // [0..1] 64-bit arrival counter, [2] failed hand-offs (sticky count), [3] grid size of the first launch
    int nq_wg, nr_wg, nf_wg, np_wg;      // workgroups per role, in block order: q_proj | r0 | fold-w | fold-pos
};

// (round 4: 2 / 4 rows per wave -- 299 workgroups, 18 / 36 KB of cold weights each -- measured 10.9 us against 9.3 us: more than
// one workgroup per CU puts consumers beside producers, as round 3 found)
#ifdef HICOM_TRACE
__device__ unsigned long long g_prep_trace[512 * 8];      // dev-only (tools/prep_trace.py): s_memrealtime stamps per workgroup of the last launch
#define PREP_TR(k) do { if (threadIdx.x == 0) g_prep_trace[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PREP_TR(k) do {} while (0)
#endif

constexpr int kQRows = 4;                // q_proj outputs per wave (72 workgroups at E = 1152: the whole grid stays <= 256 workgroups)
constexpr int kRRows = 4;                // r0 outputs per wave (round 5: 8 made the 28 r0 workgroups -- 73 KB of cold weights each -- the LAST to leave the launch, tools/prep_trace.py)
constexpr int kPrepCh = 3;               // 16-byte chunks per lane and row: K <= 1536

__device__ __forceinline__ void prep_load_x(const uint16_t* g, const uint16_t* add, int K, int lane, float (&x)[kPrepCh][8]) {
#pragma unroll
    for (int c = 0; c < kPrepCh; ++c) {
        const int k = (lane + 64 * c) * 8;
        if (k < K) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(g + k);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[c][2 * i] = bf16lo_to_f32(v[i]);
                x[c][2 * i + 1] = bf16hi_to_f32(v[i]);
            }
            if (add) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(add + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x[c][2 * i] += bf16lo_to_f32(a[i]);
                    x[c][2 * i + 1] += bf16hi_to_f32(a[i]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[c][i] = 0.f;
        }
    }
}

// The weight rows n0 .. n0 + NR - 1 of w [N, K], requested together (one memory round trip per batch) ...
template <int NR>
__device__ __forceinline__ void prep_load_w(const uint16_t* w, int N, int K, int n0, int lane, u32x4 (&wv)[NR][kPrepCh]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = n0 + r < N ? n0 + r : N - 1;
#pragma unroll
        for (int c = 0; c < kPrepCh; ++c) {
            const int k = (lane + 64 * c) * 8;
            wv[r][c] = (k < K) ? *reinterpret_cast<const u32x4*>(w + (long)n * K + k) : u32x4{0, 0, 0, 0};
        }
    }
}
// ... and their dot products with x: out[r] in all lanes
template <int NR>
__device__ __forceinline__ void prep_dot(const u32x4 (&wv)[NR][kPrepCh], const float (&x)[kPrepCh][8], float (&out)[NR]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < kPrepCh; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc = fmaf(bf16lo_to_f32(wv[r][c][i]), x[c][2 * i], acc);
                acc = fmaf(bf16hi_to_f32(wv[r][c][i]), x[c][2 * i + 1], acc);
            }
        out[r] = wave_sum_fast(acc);
    }
}

// Epoch of this launch = (arrivals counted so far) / (workgroups per launch) + 1.  Every workgroup adds ONE arrival (a
// fire-and-forget atomic, its last instruction) and reads the counter before that: at the start of launch i the counter
// holds exactly i * gridDim.x (stream order: launch i - 1 has completed), and while launch i runs it stays below
// (i + 1) * gridDim.x for every workgroup that has not yet added its own arrival -- so all workgroups of a launch derive the
// same epoch with no returning atomic, no publication step and no memset.  64-bit counter (never wraps); the grid size must be
// the same for every launch that uses one state block (it is owned by one plan).
__device__ __forceinline__ unsigned read_epoch(gu64* cnt) {
    const unsigned long long c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)(c / gridDim.x) + 1u;
}

__global__ __launch_bounds__(256) void query_prep_kernel(const uint16_t* g0, const uint16_t* wq0, const uint16_t* bq0, unsigned* state0, int E0, int nq_wg0, int nr_wg0, PrepParams p_in) {
    // (leading scalars: what the q_proj workgroups' first requests depend on -- they head the launch's critical path; preloaded into SGPRs at
    // wave launch, build_native.py -amdgpu-kernarg-preload-count)
    PrepParams p = p_in;
    p.g = g0; p.wq = wq0; p.bq = bq0; p.state = state0; p.E = E0; p.nq_wg = nq_wg0; p.nr_wg = nr_wg0;
    __shared__ float qs[128];                                        // q_proj outputs of this workgroup's head
    __shared__ __attribute__((aligned(16))) float red[16][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    gu64* cnt = (gu64*)p.state;
    gu64* gran = (gu64*)p.gran;
    PREP_TR(0);

    if (b < p.nq_wg) {
        // ---- q_proj: wave -> kQRows consecutive outputs (few rows per wave, many workgroups: the stage is one cold-memory
        // round trip, so the fewer bytes a CU pulls the sooner the granules leave) -> granules --------------------------------
        u32x4 wv[kQRows][kPrepCh];
        const int n0 = (b * 4 + wave) * kQRows;
        prep_load_w<kQRows>(p.wq, p.E, p.E, n0 < p.E ? n0 : 0, lane, wv);
        float x[kPrepCh][8];
        prep_load_x(p.g, nullptr, p.E, lane, x);
        // the epoch is needed at the store only: its (memory-side) read runs under the weight loads; so does the bias
        const unsigned epoch = read_epoch(cnt);
        const int nb = n0 + (lane < kQRows ? lane : 0);
        const float bias = (p.bq && nb < p.E) ? bf16_to_f32(p.bq[nb]) : 0.f;
        PREP_TR(1);   // q_proj: loads requested
        float out[kQRows];
        prep_dot<kQRows>(wv, x, out);
        PREP_TR(2);   // q_proj: dots done
        float v = out[0];
#pragma unroll
        for (int r = 1; r < kQRows; ++r) v = lane == r ? out[r] : v;
        const int n = n0 + lane;
        if (lane < kQRows && n < p.E) {
            v += bias;
            __hip_atomic_store(gran + n, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        PREP_TR(3);   // q_proj: granules stored
    } else if (b < p.nq_wg + p.nr_wg) {
        // ---- r0 = G0 (b_o + g) + g_b0: independent of everything else in this launch --------------------------------------
        u32x4 wv[kRRows][kPrepCh];
        const int n0 = ((b - p.nq_wg) * 4 + wave) * kRRows;
        prep_load_w<kRRows>(p.gw0, p.hidden, p.E, n0 < p.hidden ? n0 : 0, lane, wv);
        float x[kPrepCh][8];
        prep_load_x(p.g, p.bo, p.E, lane, x);
        const int nb = n0 + (lane < kRRows ? lane : 0);
        const float bias = nb < p.hidden ? bf16_to_f32(p.gb0[nb]) : 0.f;
        float out[kRRows];
        prep_dot<kRRows>(wv, x, out);
        float v = out[0];
#pragma unroll
        for (int r = 1; r < kRRows; ++r) v = lane == r ? out[r] : v;
        const int n = n0 + lane;
        if (lane < kRRows && n < p.hidden) p.r0[n] = v + bias;
    } else {
        // ---- fold roles: consumers of head h's granules -------------------------------------------------------------------
        const bool is_w = b < p.nq_wg + p.nr_wg + p.nf_wg;
        const int idx = is_w ? b - p.nq_wg - p.nr_wg : b - p.nq_wg - p.nr_wg - p.nf_wg;
        const int hd = p.hd;
        int h, slab;
        u32x4 wt[8];                // fold-w: 8 weight rows x this thread's 8 channels
        float kp[32];               // fold-pos: 32 kpe rows x this thread's position
        if (is_w) {
            const int nslab = p.E >> 7;
            h = idx / nslab; slab = idx - h * nslab;
            const int jg = tid >> 4, cl = tid & 15;
            // weights first: their (cold) latency runs under the wait for the granules
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int j = jg * 8 + r;
                wt[r] = (j < hd) ? *reinterpret_cast<const u32x4*>(p.wk + (long)(h * hd + j) * p.E + slab * 128 + cl * 8) : u32x4{0, 0, 0, 0};
            }
        } else {
            const int nslab = (p.P + 63) >> 6;
            h = idx / nslab; slab = idx - h * nslab;
            const int jg = tid >> 6, pc = slab * 64 + (tid & 63);
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int j = jg * 32 + r;
                kp[r] = (j < hd && pc < p.P) ? p.kpe[(long)(h * hd + j) * p.P + pc] : 0.f;
            }
        }
        PREP_TR(1);   // fold: weights requested
        if (wave == 0) {
            // one wave sweeps the head's granules (<= 128: two per lane) until every tag carries this launch's epoch
            const unsigned epoch = read_epoch(cnt);
            const int j0 = lane, j1 = lane + 64;
            unsigned long long a = 0, c = 0;
            unsigned spins = 0;
            // the epoch arithmetic holds only while every launch on this state block has the same grid: word 3 remembers the
            // first launch's (block 0 records it); a different grid is a failed hand-off like a spin that gave up
            const unsigned grid0 = __hip_atomic_load((gu32*)p.state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool failed = grid0 != 0u && grid0 != gridDim.x;
            for (;;) {
                a = (j0 < hd) ? __hip_atomic_load(gran + h * hd + j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)epoch << 32);
                c = (j1 < hd) ? __hip_atomic_load(gran + h * hd + j1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)epoch << 32);
                const bool ok = (unsigned)(a >> 32) == epoch && (unsigned)(c >> 32) == epoch;
                if (__all(ok)) break;
                if (failed || ++spins > (1u << 22)) {                 // give up (~seconds): never hang the queue
                    failed = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            // A missed hand-off must not pass for a result: the fold outputs of this workgroup become NaN (they reach every
            // output token through the scores) and the sticky count in state word 2 says why.
            if (failed && lane == 0) atomicAdd(p.state + 2, 1u);
            const float poison = __uint_as_float(0x7FC00000u);
            if (j0 < hd) qs[j0] = failed ? poison : __uint_as_float((unsigned)a);
            if (j1 < hd) qs[j1] = failed ? poison : __uint_as_float((unsigned)c);
        }
        __syncthreads();
        PREP_TR(2);   // fold: granules swept
        if (is_w) {
            const int jg = tid >> 4, cl = tid & 15;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float q = jg * 8 + r < hd ? qs[jg * 8 + r] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] = fmaf(bf16lo_to_f32(wt[r][i]), q, acc[2 * i]);
                    acc[2 * i + 1] = fmaf(bf16hi_to_f32(wt[r][i]), q, acc[2 * i + 1]);
                }
            }
            *reinterpret_cast<f32x4*>(&red[jg][cl * 8]) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(&red[jg][cl * 8 + 4]) = f32x4{acc[4], acc[5], acc[6], acc[7]};
            __syncthreads();
            if (tid < 64) {
                const int c = slab * 128 + 2 * tid;
                float v0 = 0.f, v1 = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < 16; ++g2) {
                    v0 += red[g2][2 * tid];
                    v1 += red[g2][2 * tid + 1];
                }
                v0 *= p.scale; v1 *= p.scale;
                uint16_t h0, l0, h1, l1;
                split_bf16(v0, h0, l0);
                split_bf16(v1, h1, l1);
                *reinterpret_cast<uint32_t*>(p.qhi + (long)h * p.E + c) = (uint32_t)h0 | ((uint32_t)h1 << 16);
                *reinterpret_cast<uint32_t*>(p.qlo + (long)h * p.E + c) = (uint32_t)l0 | ((uint32_t)l1 << 16);
            } else if (tid < 128) {
                // local query rows R .. 15 of the hi plane = the guide (lo plane stays zero): rows R + h, R + h + nh, ...
                const int c = slab * 128 + 2 * (tid - 64);
                const uint32_t g2 = *reinterpret_cast<const uint32_t*>(p.lq + c);
                for (int r = p.R + h; r < 16; r += p.nh) *reinterpret_cast<uint32_t*>(p.qhi + (long)r * p.E + c) = g2;
            }
        } else {
            const int jg = tid >> 6, pc = tid & 63;
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) acc = fmaf(kp[r], jg * 32 + r < hd ? qs[jg * 32 + r] : 0.f, acc);
            red[jg][pc] = acc;
            __syncthreads();
            const int pp = slab * 64 + tid;
            if (tid < 64 && pp < p.P) p.pos_a[(long)h * p.pos_stride + pp] = p.scale * ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
        }
    }
    // the arrival: after every wave of the workgroup has read the counter (they all did before their first barrier / store)
    __syncthreads();
    PREP_TR(7);
    if (tid == 0 && b == 0 && __hip_atomic_load((gu32*)p.state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        __hip_atomic_store((gu32*)p.state + 3, gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // result unused: no-return atomic
}

}  // namespace hicom

using namespace hicom;

#ifdef HICOM_TRACE
extern "C" int hicom_debug_prep_trace(void* dst, int64_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(hicom::g_prep_trace), (size_t)bytes) == hipSuccess ? HICOM_OK : HICOM_ELAUNCH;
}
#endif

// [state words: one 256-byte block of its own | granules].  The arrival counter takes an atomic add from every workgroup of every launch and
// an epoch read from most: round 4 had the granules start 64 bytes behind it, i.e. the first eight granules of head 0 on the counter's
// 128-byte line, and head 0's fold workgroups saw their granules 1.5-2 us after everybody else's (tools/prep_trace.py).
constexpr int kPrepStateHead = 256;
extern "C" int64_t hicom_query_prep_state_bytes(int32_t E) { return (int64_t)E * 8 + kPrepStateHead; }

extern "C" int hicom_query_prep_fwd(const void* guide, const void* local_q, const void* w_q, const void* b_q, const void* w_k, const float* kpe,
                                    int32_t nh, int32_t E, int32_t P, float scale, void* qt_hi, void* qt_lo, float* pos_a,
                                    int32_t pos_stride, int32_t rows, const void* g_w0, const void* g_b0, const void* b_o,
                                    int32_t hidden, float* r0, void* state, void* stream) {
    HICOM_REQUIRE(guide && local_q && w_q && w_k && qt_hi && qt_lo && state, HICOM_EINVAL, "query_prep: NULL pointer");
    HICOM_REQUIRE(nh > 0 && E > 0 && E % nh == 0 && E / nh <= 128 && E % 128 == 0 && E <= 1536 && rows == nh && rows <= 16, HICOM_EINVAL,
                  "query_prep: bad shape (E %% 128, head dim <= 128, one query row per head)");
    HICOM_REQUIRE(!kpe || (pos_a && P > 0 && pos_stride >= P), HICOM_EINVAL, "query_prep: positional outputs");
    HICOM_REQUIRE(!g_w0 || (g_b0 && b_o && r0 && hidden > 0), HICOM_EINVAL, "query_prep: r0 arguments");
    HICOM_REQUIRE(((uintptr_t)guide % 16 == 0) && ((uintptr_t)local_q % 4 == 0) && ((uintptr_t)w_q % 16 == 0) && ((uintptr_t)w_k % 16 == 0) && ((uintptr_t)state % 16 == 0) &&
                      (!g_w0 || ((uintptr_t)g_w0 % 16 == 0 && (uintptr_t)b_o % 16 == 0)), HICOM_EINVAL, "query_prep: 16-byte alignment");
    PrepParams p;
    p.g = (const uint16_t*)guide; p.lq = (const uint16_t*)local_q; p.wq = (const uint16_t*)w_q; p.bq = (const uint16_t*)b_q; p.wk = (const uint16_t*)w_k; p.kpe = kpe;
    p.E = E; p.nh = nh; p.hd = E / nh; p.P = kpe ? P : 0; p.scale = scale;
    p.qhi = (uint16_t*)qt_hi; p.qlo = (uint16_t*)qt_lo; p.pos_a = pos_a; p.pos_stride = pos_stride; p.R = rows;
    p.gw0 = (const uint16_t*)g_w0; p.gb0 = (const uint16_t*)g_b0; p.bo = (const uint16_t*)b_o; p.hidden = g_w0 ? hidden : 0; p.r0 = r0;
    p.gran = (unsigned long long*)((char*)state + kPrepStateHead); p.state = (unsigned*)state;
    p.nq_wg = (E + 4 * kQRows - 1) / (4 * kQRows);
    p.nr_wg = p.hidden ? (p.hidden + 4 * kRRows - 1) / (4 * kRRows) : 0;
    p.nf_wg = nh * (E / 128);
    p.np_wg = p.P ? nh * ((p.P + 63) / 64) : 0;
    const int grid = p.nq_wg + p.nr_wg + p.nf_wg + p.np_wg;
    HICOM_LAUNCH(query_prep_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p.g, p.wq, p.bq, p.state, p.E, p.nq_wg, p.nr_wg, p);
    return hicom_host::check_launch("query_prep");
}
