"""Helpers for the CPU-emulation tests: build/load libspr_emu.so and make a scorer on it."""

import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


@functools.lru_cache(maxsize=None)
def emu_library():
    import build_emu  # tests/emu/build_emu.py
    from shoeprint_image_retrieval_amd import _lib

    return _lib.load_library(build_emu.build())


def emu_scorer(method="auto", **kw):
    from host_device import HostDevice
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    return NccScorer(device=HostDevice(), library=emu_library(), method=method, **kw)
