// Downstream neighbour of the compressor: splicing the compressed visual tokens into the LLM's input
// embeddings at the <image> / <video> placeholders (reference hicom/model/hicom_arch.py:271-373,
// prepare_inputs_labels_for_multimodal).  The reference does it with embed_tokens calls on id slices, torch.cat per
// sample, a zero-pad cat and a stack -- every embedding row is copied three times.  Here the host plans the row layout
// once (integer work on the tiny id tensor) and ONE kernel writes each output row exactly once from its source:
//   a row of the embedding table, a row of a compressed-token tensor, or zeros (right padding).
// A second small kernel builds the labels / attention mask of the new layout.
#include "common.hpp"

namespace hicom {

// dst[r, :] = *(row_src[r]) (row_bytes bytes, 16-byte vectors) or zeros when row_src[r] == 0.  One wave per row.
__global__ __launch_bounds__(256) void splice_rows_kernel(const unsigned long long* row_src, long nrows, int row_bytes, char* dst) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    const char* s = reinterpret_cast<const char*>(row_src[r]);
    char* d = dst + r * (long)row_bytes;
    for (int c = lane * 16; c < row_bytes; c += 64 * 16)
        *reinterpret_cast<u32x4*>(d + c) = s ? *reinterpret_cast<const u32x4*>(s + c) : u32x4{0, 0, 0, 0};
}

// new_labels[b, p] = labels[b, map[b, p]] (map >= 0) or IGNORE (visual tokens, padding)                (:309-311,:344-348)
// new_mask[b, p]   = 1 for p < L_b - S | mask[b, p - (L_b - S)] for p < L_b | 0 (right padding)          (:350-366)
// mask elements are msz bytes wide (1: torch.bool, 8: torch.long), copied verbatim / written as 0 or 1.
__global__ __launch_bounds__(256) void splice_labels_kernel(const long* labels, const char* mask, int msz, const int* map, const int* new_len,
                                                            int B, int S, int Lmax, long ignore, long* new_labels, char* new_mask) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * Lmax) return;
    const int b = (int)(i / Lmax), p = (int)(i - (long)b * Lmax);
    if (new_labels) {
        const int m = map[i];
        new_labels[i] = (labels && m >= 0) ? labels[(long)b * S + m] : ignore;
    }
    if (new_mask) {
        const int L = new_len[b], left = L - S;
        char* o = new_mask + i * msz;
        for (int k = 0; k < msz; ++k) o[k] = 0;
        if (p < left) o[0] = 1;
        else if (p < L)
            for (int k = 0; k < msz; ++k) o[k] = mask[((long)b * S + (p - left)) * msz + k];
    }
}

}  // namespace hicom

using namespace hicom;

extern "C" int hicom_splice_rows_fwd(const void* row_src, int64_t nrows, int32_t row_bytes, void* dst, void* stream) {
    HICOM_REQUIRE(row_src && dst && nrows > 0 && row_bytes > 0 && row_bytes % 16 == 0 && (uintptr_t)dst % 16 == 0, HICOM_EINVAL,
                  "splice_rows: bad arguments (rows of a multiple of 16 bytes)");
    hipLaunchKernelGGL(splice_rows_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)row_src, (long)nrows, row_bytes, (char*)dst);
    return hicom_host::check_launch("splice_rows");
}

extern "C" int hicom_splice_labels_fwd(const void* labels, const void* mask, int32_t mask_elem_bytes, const int32_t* map,
                                       const int32_t* new_len, int32_t B, int32_t S, int32_t Lmax, int64_t ignore_index,
                                       void* new_labels, void* new_mask, void* stream) {
    HICOM_REQUIRE(map && new_len && B > 0 && S > 0 && Lmax > 0 && (new_labels || new_mask), HICOM_EINVAL, "splice_labels: bad arguments");
    HICOM_REQUIRE(!new_mask || (mask && (mask_elem_bytes == 1 || mask_elem_bytes == 8)), HICOM_EINVAL,
                  "splice_labels: attention mask of torch.bool or torch.long");
    const long n = (long)B * Lmax;
    hipLaunchKernelGGL(splice_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long*)labels,
                       (const char*)mask, mask_elem_bytes, map, new_len, B, S, Lmax, (long)ignore_index, (long*)new_labels, (char*)new_mask);
    return hicom_host::check_launch("splice_labels");
}
