"""Host-memory backend for driving the CPU-emulation build through the product's host code
(test infrastructure: "device" buffers are numpy arrays, pointers are host pointers)."""

from __future__ import annotations

import numpy as np


class HostDevice:
    name = "emu"

    def empty(self, shape, dtype=np.float32):
        return np.full(tuple(int(s) for s in np.atleast_1d(shape)), 0xCD, dtype=np.uint8).view(np.uint8)[: 0] if False else \
            np.empty(tuple(int(s) for s in np.atleast_1d(shape)), dtype=dtype)

    def empty_bytes(self, nbytes: int):
        # poison so that reads of unwritten prepared data are visible
        return np.full(int(nbytes), 0xCD, dtype=np.uint8)

    def zeros(self, shape, dtype=np.float32):
        return np.zeros(tuple(int(s) for s in np.atleast_1d(shape)), dtype=dtype)

    def to_device(self, array):
        return np.array(array, copy=True, order="C")

    def stack_to_device(self, items):
        return np.ascontiguousarray(np.stack([np.asarray(i, dtype=np.float32) for i in items]))

    def channel(self, buf, k):
        return np.ascontiguousarray(buf[..., k])

    def set_channel(self, buf, k, plane):
        buf[..., k] = plane

    def astype_storage(self, buf, storage: str):
        if storage == "float32":
            return buf
        if storage == "float16":
            return np.ascontiguousarray(buf.astype(np.float16))
        bits = np.ascontiguousarray(buf, dtype=np.float32).view(np.uint32).astype(np.uint64)  # bfloat16, round to nearest even
        return (((bits + 0x7FFF + ((bits >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)

    def to_host(self, buf):
        return np.array(buf, copy=True)

    def is_device_array(self, obj):
        return False

    def ptr(self, buf) -> int:
        assert buf.flags["C_CONTIGUOUS"]
        return int(buf.ctypes.data)

    def stream(self) -> int:
        return 0

    def synchronize(self):
        pass

    # streams: the emulation runs every call to completion, so they are all the one host "stream"
    def new_stream(self):
        return None

    def current(self):
        return None

    def use(self, stream):
        import contextlib

        return contextlib.nullcontext()

    def record_event(self, stream):
        return None

    def wait_event(self, stream, event):
        pass

    def wait_stream(self, waiter, waited):
        pass

    def record_stream(self, buf, stream):
        pass

    def free_bytes(self) -> int:
        return 8 << 30

    def shape(self, buf):
        return tuple(buf.shape)

    def narrow0(self, buf, start, length):
        return buf[int(start): int(start) + int(length)]
