"""Device-memory plumbing for the host mirror: PyTorch-ROCm owns HBM allocations and streams.

``TorchDevice`` is the only backend the product uses.  It refuses to construct without a
GPU: there is no CPU execution path in the product (the tests drive the same host logic
against the CPU-emulation build through ``tests/emu/host_device.py``).
"""

from __future__ import annotations

import numpy as np

_NP_TO_TORCH = {}


class TorchDevice:
    """HBM buffers as torch tensors on one GPU; work is enqueued on torch's current stream."""

    name = "hip"

    def __init__(self, device: str | int | None = None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError(
                "No ROCm GPU visible: the shoeprint scorer runs only on the HIP path "
                "(there is deliberately no CPU fallback)."
            )
        self.torch = torch
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)

    # -- allocation -------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float32):
        return self.torch.empty(tuple(int(s) for s in np.atleast_1d(shape)), dtype=self._dtype(dtype), device=self.device)

    def empty_bytes(self, nbytes: int):
        return self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)

    def zeros(self, shape, dtype=np.float32):
        return self.torch.zeros(tuple(int(s) for s in np.atleast_1d(shape)), dtype=self._dtype(dtype), device=self.device)

    def _dtype(self, dtype):
        t = self.torch
        table = {np.dtype(np.float32): t.float32, np.dtype(np.int32): t.int32, np.dtype(np.uint8): t.uint8,
                 np.dtype(np.float16): t.float16, np.dtype(np.int64): t.int64}
        return table[np.dtype(dtype)]

    # -- transfers --------------------------------------------------------------------------
    def to_device(self, array: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(array)).to(self.device, non_blocking=False)

    def stack_to_device(self, items):
        """float32 device batch [N, ...] from a list of equally shaped host arrays, copied item by item straight
        into the device tensor (no stacked host copy first: the host pass is the slow part of this boundary)."""
        first = np.asarray(items[0])
        out = self.torch.empty((len(items),) + tuple(first.shape), dtype=self.torch.float32, device=self.device)
        for k, it in enumerate(items):
            out[k].copy_(self.torch.from_numpy(np.ascontiguousarray(it, dtype=np.float32)))
        return out

    def channel(self, buf, k: int):
        """Contiguous copy of plane k of an interleaved [..., C] buffer."""
        return buf[..., k].contiguous()

    def set_channel(self, buf, k: int, plane) -> None:
        buf[..., k].copy_(plane)

    def astype_storage(self, buf, storage: str):
        """The float32 batch in its HBM storage type ("float32" | "float16" | "bfloat16"); the kernels read all three."""
        t = {"float32": self.torch.float32, "float16": self.torch.float16, "bfloat16": self.torch.bfloat16}[storage]
        return buf if buf.dtype == t else buf.to(t)

    def to_host(self, buf) -> np.ndarray:
        return buf.detach().cpu().numpy()

    def is_device_array(self, obj) -> bool:
        return isinstance(obj, self.torch.Tensor) and obj.is_cuda

    # -- raw handles for the C ABI ----------------------------------------------------------
    def ptr(self, buf) -> int:
        return int(buf.data_ptr())

    def stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.device)

    # -- streams (multi-layer pipeline: extractor and per-layer scoring chains overlap) ----------------
    def new_stream(self):
        return self.torch.cuda.Stream(device=self.device)

    def current(self):
        return self.torch.cuda.current_stream(self.device)

    def use(self, stream):
        """Context manager: work enqueued inside goes to ``stream`` (the C-ABI calls take torch's current stream)."""
        return self.torch.cuda.stream(stream)

    def record_event(self, stream):
        ev = self.torch.cuda.Event()
        ev.record(stream)
        return ev

    def wait_event(self, stream, event) -> None:
        stream.wait_event(event)

    def wait_stream(self, waiter, waited) -> None:
        waiter.wait_stream(waited)

    def record_stream(self, buf, stream) -> None:
        """``buf`` (allocated on another stream) is used by work enqueued on ``stream``: its memory must not be handed out
        again before that work has run."""
        buf.record_stream(stream)

    def free_bytes(self) -> int:
        free, _total = self.torch.cuda.mem_get_info(self.device)
        return int(free)

    def shape(self, buf):
        return tuple(buf.shape)

    def narrow0(self, buf, start: int, length: int):
        return buf.narrow(0, int(start), int(length))
