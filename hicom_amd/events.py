"""Events that order two streams of ONE device.

`torch.cuda.Event()` creates its HIP event with hipEventDisableTiming only; when such an event completes the runtime performs a SYSTEM-scope
fence in front of the next launch of the recording stream -- measured on the frame-sharded step as a 5-6-us gap between readout GEMM 2 and
the next step's query prep (tools/shard_trace.py; pipelined 80-82 -> 78-79 us without it, profiles/r06_j_shard_events.txt).  Ordering a side
or comm stream of the same device behind the caller's stream needs device scope only: these events add hipEventDisableSystemFence, which
torch's constructor cannot ask for.  Events a CALLER waits on from the host or hands to other libraries stay torch events.
HICOM_EVENT_NOFENCE=0: torch events everywhere (A/B switch)."""
from __future__ import annotations

import ctypes
import os

import torch

_HIP = None
_DISABLE_TIMING, _DISABLE_SYSTEM_FENCE = 0x2, 0x20000000


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")       # (the runtime torch has loaded, by soname)
        _HIP.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        _HIP.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _HIP.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        _HIP.hipEventDestroy.argtypes = [ctypes.c_void_p]
    return _HIP


class DeviceEvent:
    """hipEventDisableTiming | hipEventDisableSystemFence; `.cuda_event` is the raw handle the C ABI takes."""
    __slots__ = ("cuda_event",)

    def __init__(self):
        h = ctypes.c_void_p()
        rc = _hip().hipEventCreateWithFlags(ctypes.byref(h), _DISABLE_TIMING | _DISABLE_SYSTEM_FENCE)
        if rc != 0 or not h.value:
            raise RuntimeError(f"hipEventCreateWithFlags failed ({rc})")
        self.cuda_event = h.value

    def record(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = _hip().hipEventRecord(self.cuda_event, s.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"hipEventRecord failed ({rc})")

    def wait(self, stream=None):
        """`stream` (default: the current one) waits for the event."""
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = _hip().hipStreamWaitEvent(s.cuda_stream, self.cuda_event, 0)
        if rc != 0:
            raise RuntimeError(f"hipStreamWaitEvent failed ({rc})")

    def __del__(self):
        try:
            _HIP.hipEventDestroy(self.cuda_event)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def device_event():
    """A DeviceEvent, or -- HICOM_EVENT_NOFENCE=0, or no libamdhip64 under that name -- a torch event (same record / wait / cuda_event surface)."""
    if os.environ.get("HICOM_EVENT_NOFENCE", "1") != "0":
        try:
            return DeviceEvent()
        except Exception:  # noqa: BLE001
            pass
    return torch.cuda.Event()
