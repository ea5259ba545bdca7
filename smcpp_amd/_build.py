"""Builds ``libsmcpp_engine.so`` (HIP kernels + C ABI) in-tree for gfx950.

``hipcc`` cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting ``.so`` is
git-ignored and travels to the GPU box with the working tree."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsmcpp_engine.so")
SOURCES = ["engine.hip"]
DEPS = ["engine.hip", "kernels.hpp", "nonsym_eig.hpp", "prep.hpp", os.path.join("..", "..", "include", "smcpp_engine.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fopenmp", "-fPIC", "-shared",
           "-Wl,-rpath,/opt/rocm/lib/llvm/lib", "-Wl,-rpath,/opt/rocm/lib",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
