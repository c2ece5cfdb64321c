"""``torch.ops.shoeprint_mi355x`` — the PyTorch-ROCm custom-op face of the C ABI (csrc/torch_ops.cpp).

``load()`` registers the operators (``TORCH_LIBRARY``) by loading ``libshoeprint_torch_ops.so`` (built in-tree next to
``libshoeprint_mi355x.so``, which it links) and returns the op namespace:

* ``ncc_scores(q, g, crop=2, method="auto", max_prepared_bytes=0) -> float32 [Q, G]``   (similarity.py:129-227)
* ``ranks(scores, match) -> int32 [Q]``                                                (similarity.py:378-386)
* ``extract(images, packed, arch, block, mean, std) -> float32 [N, C, h, w]``          (network.py:210-244, plain VGG)

They take GPU tensors, run on PyTorch's current HIP stream and never synchronise; the ctypes route (``_lib.py``) calls the
very same entry points.  ``SPR_TORCH_OPS=0`` keeps the host mirror on the ctypes route (A/B and parity tests).
"""

from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "libshoeprint_torch_ops.so")
_ops = None


def load():
    """The op namespace; raises if the extension is not built (``make -C shoeprint-image-retrieval_amd/csrc``)."""
    global _ops
    if _ops is None:
        import torch

        if not os.path.exists(PATH):
            raise RuntimeError(f"{PATH} not found: build it with `make -C shoeprint-image-retrieval_amd/csrc`")
        torch.ops.load_library(PATH)
        _ops = torch.ops.shoeprint_mi355x
    return _ops


def enabled() -> bool:
    """Should the host mirror route through torch.ops?  Yes when the extension is built and not switched off."""
    return os.environ.get("SPR_TORCH_OPS", "1") != "0" and os.path.exists(PATH)
