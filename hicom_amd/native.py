"""ctypes binding of libhicom_hip.so (the C ABI declared in include/hicom_hip.h).

The HIP library is the product: there is NO CPU or PyTorch fallback.  `lib()` raises if the
shared object is missing or was built for a different ABI version; every wrapper raises
`HicomNativeError` on a non-zero status, with the library's own message.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhicom_hip.so")
ABI_VERSION = 1

DT_BF16, DT_F32 = 0, 1
ACT_NONE, ACT_GELU = 0, 1

EXPORTS = (
    "hicom_abi_version", "hicom_last_error", "hicom_local_attn_fwd", "hicom_trilinear_pool_fwd",
    "hicom_linear_fwd", "hicom_fold_query_fwd", "hicom_split_bf16_fwd", "hicom_global_stream_fwd",
    "hicom_global_stream_nparts", "hicom_global_merge_fwd", "hicom_global_combine_fwd",
    "hicom_readout_gemm_fwd", "hicom_scatter_rows_fwd",
)


class HicomNativeError(RuntimeError):
    pass


class Axis(C.Structure):
    """hicom_axis: one axis of the window tiling (include/hicom_hip.h)."""
    _fields_ = [("n", C.c_int32), ("k", C.c_int32), ("nwin", C.c_int32), ("nfull", C.c_int32)]


_LIB: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise HicomNativeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  hicom_amd has no non-HIP execution path.")
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise HicomNativeError(f"{LIB_PATH} does not export {name}; rebuild it")
    L.hicom_abi_version.restype = C.c_int
    L.hicom_last_error.restype = C.c_char_p
    if L.hicom_abi_version() != ABI_VERSION:
        raise HicomNativeError(f"ABI mismatch: library {L.hicom_abi_version()} vs binding {ABI_VERSION}; rebuild")
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.hicom_local_attn_fwd.argtypes = [vp, vp, i32, Axis, Axis, Axis, vp, i32, i64, f32, f32, i32, vp, vp]
    L.hicom_trilinear_pool_fwd.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.hicom_linear_fwd.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.hicom_fold_query_fwd.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp]
    L.hicom_split_bf16_fwd.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.hicom_global_stream_fwd.argtypes = [vp, i64, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, i64,
                                          vp, vp, vp, i32, vp]
    L.hicom_global_stream_nparts.argtypes = [i64, i32]
    L.hicom_global_merge_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, i64, i64, i32, i32, vp, i32, i32, i32,
                                         vp, vp, vp, vp]
    L.hicom_global_combine_fwd.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.hicom_readout_gemm_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i64, i64, i32, vp]
    L.hicom_scatter_rows_fwd.argtypes = [vp, i32, i32, i32, vp, i32, i64, i64, i64, i32, i32, vp]
    for name in EXPORTS[2:]:
        getattr(L, name).restype = C.c_int
    _LIB = L
    return L


def _check(status: int, what: str):
    if status != 0:
        msg = lib().hicom_last_error().decode(errors="replace")
        raise HicomNativeError(f"{what} failed ({status}): {msg}")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise HicomNativeError("hicom_amd kernels need device tensors (HIP); got a CPU tensor")
    if not t.is_contiguous():
        raise HicomNativeError("hicom_amd kernels need dense row-major tensors")
    return C.c_void_p(t.data_ptr())


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float32:
        return DT_F32
    raise HicomNativeError(f"unsupported dtype {t.dtype} (bf16 or f32 only)")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------
# thin typed wrappers (one per entry point)
# ------------------------------------------------------------------------------------------
def local_attn(key, value, axes, query, query_stride, scale, bias, l2norm, ctx):
    D = value.shape[-1]
    _check(lib().hicom_local_attn_fwd(_ptr(key), _ptr(value), D, axes[0], axes[1], axes[2], _ptr(query), _dt(query),
                                      query_stride, scale, bias, l2norm, _ptr(ctx), _stream()), "hicom_local_attn_fwd")


def trilinear_pool(x, out):
    T, H, W, D = x.shape
    To, Ho, Wo, _ = out.shape
    _check(lib().hicom_trilinear_pool_fwd(_ptr(x), T, H, W, D, To, Ho, Wo, _ptr(out), _stream()),
           "hicom_trilinear_pool_fwd")


def linear(x, w, b, y, res=None, res_bcast=False, act=ACT_NONE, head_rows=0, head_dim=0, M=None):
    N, K = w.shape
    M = y.shape[0] if M is None else M
    _check(lib().hicom_linear_fwd(_ptr(x), _dt(x), _ptr(w), _dt(w), _ptr(b), _dt(b) if b is not None else 0,
                                  _ptr(res), int(res_bcast), M, N, K, head_rows, head_dim, act, _ptr(y), _stream()),
           "hicom_linear_fwd")


def fold_query(qp, w_k, nh, scale, qt):
    nq, E = qp.shape
    _check(lib().hicom_fold_query_fwd(_ptr(qp), _ptr(w_k), nq, nh, E, scale, _ptr(qt), _stream()), "hicom_fold_query_fwd")


def split_bf16(x, rows_pad, hi, lo):
    rows, E = x.shape
    _check(lib().hicom_split_bf16_fwd(_ptr(x), rows, rows_pad, E, _ptr(hi), _ptr(lo), _stream()), "hicom_split_bf16_fwd")


def global_stream_nparts(N, rows_pad) -> int:
    n = lib().hicom_global_stream_nparts(N, rows_pad)
    if n <= 0:
        raise HicomNativeError("hicom_global_stream_nparts: bad arguments")
    return n


def global_stream(x, N, qhi, qlo, pos_a, H, W, t0i, y0i, x0i, scores, part_m, part_l, part_acc):
    E = x.shape[-1]
    rows_pad = qhi.shape[0]
    nparts = part_m.shape[0]
    _check(lib().hicom_global_stream_fwd(_ptr(x), N, E, _ptr(qhi), _ptr(qlo), rows_pad, _ptr(pos_a),
                                         pos_a.shape[1] if pos_a is not None else 0, H, W, t0i, y0i, x0i,
                                         _ptr(scores), scores.shape[1], _ptr(part_m), _ptr(part_l), _ptr(part_acc),
                                         nparts, _stream()), "hicom_global_stream_fwd")


def global_merge(part_m, part_l, part_acc, rows, scores, N, H, W, pe, t0i, y0i, x0i, scratch, out_ml, out_acc):
    nparts, rows_pad = part_m.shape
    E = part_acc.shape[-1]
    _check(lib().hicom_global_merge_fwd(_ptr(part_m), _ptr(part_l), _ptr(part_acc), nparts, rows, rows_pad, E,
                                        _ptr(scores), scores.shape[1], N, H, W, _ptr(pe), t0i, y0i, x0i,
                                        _ptr(scratch), _ptr(out_ml), _ptr(out_acc), _stream()), "hicom_global_merge_fwd")


def global_combine(ml, acc, ctx):
    nsets, rows, E = acc.shape
    _check(lib().hicom_global_combine_fwd(_ptr(ml), _ptr(acc), nsets, rows, E, _ptr(ctx), _stream()),
           "hicom_global_combine_fwd")


def readout_gemm(x, w, b, y, act=ACT_NONE, row0=0, nl_group=0, M=None):
    N, K = w.shape
    M = x.shape[0] if M is None else M
    _check(lib().hicom_readout_gemm_fwd(_ptr(x), _ptr(w), _ptr(b), _dt(b) if b is not None else 0, M, N, K, act,
                                        _ptr(y), _dt(y), y.shape[-1], row0, nl_group, _stream()), "hicom_readout_gemm_fwd")


def scatter_rows(src, dst, row0, count, row_step=1, nl_group=0):
    src2 = src.reshape(-1, src.shape[-1])
    _check(lib().hicom_scatter_rows_fwd(_ptr(src2), _dt(src2), src2.shape[0], src2.shape[1], _ptr(dst), _dt(dst),
                                        dst.shape[-1], row0, row_step, nl_group, count, _stream()), "hicom_scatter_rows_fwd")
