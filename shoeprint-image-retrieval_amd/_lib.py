"""ctypes binding of libshoeprint_mi355x.so (the C ABI in include/shoeprint_mi355x.h).

The library is the product: there is no Python/CPU fallback.  ``load_library()`` raises if
the in-tree shared object is missing (run ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C shoeprint-image-retrieval_amd/csrc``).
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "libshoeprint_mi355x.so")

SPR_OK = 0
F32, F16, BF16 = 0, 1, 2
NCC_AUTO, NCC_FFT, NCC_DIRECT, NCC_FFT_POW2, NCC_MFMA = 0, 1, 2, 3, 4
METHOD_NAMES = {NCC_AUTO: "auto", NCC_FFT: "fft", NCC_DIRECT: "direct", NCC_FFT_POW2: "fft_pow2", NCC_MFMA: "mfma"}


class NccShape(C.Structure):
    _fields_ = [
        ("channels", C.c_int32),
        ("q_h", C.c_int32), ("q_w", C.c_int32),
        ("g_h", C.c_int32), ("g_w", C.c_int32),
        ("crop", C.c_int32),
        ("dtype", C.c_int32),
        ("method", C.c_int32),
    ]


class SprError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libshoeprint_mi355x error {code}: {message}")
        self.code = code


# name -> (restype, argtypes); every symbol include/shoeprint_mi355x.h declares
_VP, _I64, _I32, _SZ = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
SIGNATURES = {
    "spr_last_error": (C.c_char_p, []),
    "spr_abi_version": (C.c_int, []),
    "spr_ncc_plan_create": (C.c_int, [C.POINTER(NccShape), C.POINTER(_VP)]),
    "spr_ncc_plan_destroy": (None, [_VP]),
    "spr_ncc_plan_method": (C.c_int, [_VP]),
    "spr_ncc_plan_fft_size": (C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I32)]),
    "spr_ncc_query_bytes": (_SZ, [_VP, _I64]),
    "spr_ncc_gallery_bytes": (_SZ, [_VP, _I64]),
    "spr_ncc_prepare_queries": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "spr_ncc_prepare_gallery": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "spr_ncc_score": (C.c_int, [_VP, _VP, _I64, _VP, _I64, _VP, _I64, _I64, C.c_int, _VP]),
    "spr_ncc_maps": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "spr_rank_true_match": (C.c_int, [_VP, _I64, _I64, _I64, _VP, _VP, _VP]),
    "spr_rank_count_greater": (C.c_int, [_VP, _I64, _I64, _I64, _I64, _VP, _VP, _VP, _VP]),
    "spr_scores_fuse": (C.c_int, [_VP, _VP, _I64, C.c_float, C.c_float, _VP]),
    "spr_rotate_nearest": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, C.POINTER(_I64), _VP]),
    "spr_resample_axis": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, _I32, _VP, _VP, _I32, _VP]),
    "spr_clahe_workspace_bytes": (_SZ, [_I64, _I32, _I32]),
    "spr_clahe_u8": (C.c_int, [_VP, _VP, _I64, _I32, _I32, C.c_float, _I32, _I32, _VP, _VP]),
    "spr_color_tables_bytes": (_SZ, []),
    "spr_rgb_to_lab_u8": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "spr_lab_to_rgb_u8": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "spr_vgg_plan_create": (C.c_int, [_I32, _I32, C.POINTER(_VP)]),
    "spr_vgg_plan_create_ex": (C.c_int, [_I32, _I32, _I32, C.POINTER(_VP)]),
    "spr_vgg_plan_compute": (C.c_int, [_VP]),
    "spr_vgg_conv_info": (C.c_int, [_VP, _I32, C.POINTER(_I32), C.POINTER(_I32)]),
    "spr_vgg16_plan_create": (C.c_int, [_I32, C.POINTER(_VP)]),
    "spr_vgg16_plan_destroy": (None, [_VP]),
    "spr_vgg16_num_convs": (C.c_int, [_VP]),
    "spr_vgg16_conv_shape": (C.c_int, [_VP, _I32, C.POINTER(_I32), C.POINTER(_I32)]),
    "spr_vgg16_output_shape": (C.c_int, [_VP, _I32, _I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "spr_vgg16_packed_bytes": (_SZ, [_VP]),
    "spr_vgg16_pack_weights": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), _VP, _VP]),
    "spr_vgg16_workspace_bytes": (_SZ, [_VP, _I64, _I32, _I32]),
    "spr_vgg16_forward": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    _VP, _VP, _VP, _VP]),
    "spr_vgg16_forward_taps": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         _VP, _VP, _VP, _I32, C.POINTER(_I32), C.POINTER(_VP), _VP]),
    "spr_densenet_plan_create": (C.c_int, [_I32, C.POINTER(_VP)]),
    "spr_densenet_plan_destroy": (None, [_VP]),
    "spr_densenet_num_ops": (C.c_int, [_VP]),
    "spr_densenet_op_info": (C.c_int, [_VP, _I32, C.POINTER(_I32)]),
    "spr_densenet_output_shape": (C.c_int, [_VP, _I32, _I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "spr_densenet_packed_bytes": (_SZ, [_VP]),
    "spr_densenet_workspace_bytes": (_SZ, [_VP, _I64, _I32, _I32]),
    "spr_densenet_forward": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       _VP, _VP, _VP, _VP]),
    "spr_effnet_plan_create": (C.c_int, [_I32, _I32, C.POINTER(_VP)]),
    "spr_effnet_plan_create_ex": (C.c_int, [_I32, _I32, _I32, C.POINTER(_VP)]),
    "spr_effnet_plan_compute": (C.c_int, [_VP]),
    "spr_effnet_plan_destroy": (None, [_VP]),
    "spr_effnet_num_ops": (C.c_int, [_VP]),
    "spr_effnet_op_info": (C.c_int, [_VP, _I32, C.POINTER(_I32)]),
    "spr_effnet_output_shape": (C.c_int, [_VP, _I32, _I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "spr_effnet_packed_bytes": (_SZ, [_VP]),
    "spr_effnet_workspace_bytes": (_SZ, [_VP, _I64, _I32, _I32]),
    "spr_effnet_forward": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     _VP, _VP, _VP, _VP]),
    "spr_resnet_plan_create": (C.c_int, [_I32, C.POINTER(_VP)]),
    "spr_resnet_plan_create_ex": (C.c_int, [_I32, _I32, C.POINTER(_VP)]),
    "spr_resnet_plan_compute": (C.c_int, [_VP]),
    "spr_resnet_plan_destroy": (None, [_VP]),
    "spr_resnet_num_convs": (C.c_int, [_VP]),
    "spr_resnet_conv_shape": (C.c_int, [_VP, _I32] + [C.POINTER(_I32)] * 5),
    "spr_resnet_output_shape": (C.c_int, [_VP, _I32, _I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "spr_resnet_packed_bytes": (_SZ, [_VP]),
    "spr_resnet_pack_weights": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), _VP, _VP]),
    "spr_resnet_workspace_bytes": (_SZ, [_VP, _I64, _I32, _I32]),
    "spr_resnet_forward": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     _VP, _VP, _VP, _VP]),
    "spr_synth_gallery": (C.c_int, [_VP, _I64, _I64, _I32, _I32, _I32, C.c_uint64, _VP]),
    "spr_synth_queries": (C.c_int, [_VP, _I64, _I64, _VP, _I32, _I32, _I32, C.c_uint64, _I32, _I32, _I32, _VP]),
}


class Library:
    """The loaded shared object with typed entry points; ``check`` turns status codes into SprError."""

    def __init__(self, path: str | None = None):
        self.path = path or DEFAULT_PATH
        if not os.path.exists(self.path):
            raise RuntimeError(
                f"{self.path} not found: the HIP library is the product and has no fallback. "
                "Build it with `make -C shoeprint-image-retrieval_amd/csrc` (or __graft_entry__.build())."
            )
        # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Device memory and
        # streams come from torch, so its HIP runtime must be THE runtime of the process: import torch
        # first, then our NEEDED libamdhip64.so.7 resolves to the copy that is already loaded.
        import torch  # noqa: F401

        self.cdll = C.CDLL(self.path)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the ABI is incomplete
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)
        if self.spr_abi_version() != 1:
            raise RuntimeError(f"{self.path}: ABI version {self.spr_abi_version()} != 1")

    def check(self, code: int) -> None:
        if code != SPR_OK:
            raise SprError(code, (self.spr_last_error() or b"").decode("utf-8", "replace"))


_default: Library | None = None


def load_library(path: str | None = None) -> Library:
    """The process-wide library (in-tree build).  An explicit ``path`` loads another build
    (the tests use this for the CPU-emulation twin); it never replaces the default."""
    global _default
    if path is not None:
        return Library(path)
    if _default is None:
        _default = Library()
    return _default
