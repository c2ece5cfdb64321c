#!/usr/bin/env python3
"""Build the CPU-emulation twin of libshoeprint_mi355x.so (test infrastructure).

The same .hip sources are compiled by the host clang++ against tests/emu/hip/hip_runtime.h
and tests/emu/spr_intrinsics.h, giving tests/emu/_build/libspr_emu.so with the same C ABI
("device" pointers are host pointers).  Used by the ``not gpu`` tests to execute the kernels
on tiny shapes, optionally under UBSan (SPR_EMU_SANITIZE=1).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "shoeprint-image-retrieval_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
CLANG = os.environ.get("SPR_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(sanitize: bool | None = None, verbose: bool = False) -> str:
    if sanitize is None:
        sanitize = os.environ.get("SPR_EMU_SANITIZE", "0") == "1"
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "libspr_emu_san.so" if sanitize else "libspr_emu.so")
    srcs = [os.path.join(CSRC, f) for f in sources()]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "spr_intrinsics.h"),
             os.path.join(ROOT, "include", "shoeprint_mi355x.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    flags = ["-std=c++17", "-O2", "-g", "-fPIC", "-pthread", "-I", HERE, "-I", CSRC, "-Wall",
             "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-variable", "-DSPR_EMU"]
    if sanitize:
        # trap mode: no sanitizer runtime has to be linked into (or preloaded for) the shared object; any
        # undefined behaviour executes a trap instruction and kills the test process
        flags += ["-fsanitize=undefined", "-fsanitize-trap=undefined", "-O1", "-DSPR_SAN_SUBSET"]
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT_DIR, os.path.basename(s) + (".san.o" if sanitize else ".o"))
        objs.append(o)
        cmd = [CLANG, "-x", "c++", *flags, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for s, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {s}\n{log}\n")
        elif verbose and log.strip():
            print(log)
    if failed:
        raise RuntimeError("emulation build failed")
    link = [CLANG, "-shared", "-pthread", *objs, "-o", out]
    subprocess.run(link, check=True)
    return out


if __name__ == "__main__":
    print(build(verbose=True))
