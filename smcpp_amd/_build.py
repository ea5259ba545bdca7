"""Builds ``libsmcpp_engine.so`` (HIP kernels + C ABI) in-tree for gfx950.

``hipcc`` cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting ``.so`` is
git-ignored and travels to the GPU box with the working tree."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsmcpp_engine.so")
SOURCES = ["engine.hip"]


def _deps():
    """Every source the library is compiled from: csrc/*.hip, csrc/*.hpp and the public header."""
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))] + \
           [os.path.join(HERE, "..", "include", "smcpp_engine.h"), os.path.abspath(__file__)]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def _gmp():
    """GMP's C API backs the exact-rational tables of the conditioned SFS (csrc/prep.hpp).  The image ships the runtime
    library system-wide and the header with conda; a private include directory keeps the rest of conda's headers
    off the include path."""
    inc = os.path.join(CSRC, "_gmp")
    hdr = os.path.join(inc, "gmp.h")
    if not os.path.exists(hdr):
        for cand in ("/usr/include/gmp.h", "/usr/include/x86_64-linux-gnu/gmp.h", "/opt/conda/include/gmp.h"):
            if os.path.exists(cand):
                os.makedirs(inc, exist_ok=True)
                if os.path.lexists(hdr):
                    os.remove(hdr)
                os.symlink(cand, hdr)
                break
        else:
            raise RuntimeError("gmp.h not found")
    for cand in ("/usr/lib/x86_64-linux-gnu/libgmp.so", "/usr/lib/x86_64-linux-gnu/libgmp.so.10",
                 "/opt/conda/lib/libgmp.so"):
        if os.path.exists(cand):
            return inc, cand
    raise RuntimeError("libgmp not found")


LAST_ACTION = None       # "compiled" / "reused": what the last build() call did


def build(force: bool = False, verbose: bool = False) -> str:
    global LAST_ACTION
    if not force and not _stale():
        LAST_ACTION = "reused"
        if verbose:
            print(f"smcpp_amd._build: reused {os.path.relpath(LIB)} (newer than every source under csrc/ and include/)", file=sys.stderr, flush=True)
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    gmp_inc, gmp_lib = _gmp()
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fopenmp", "-fPIC", "-shared",
           "-I" + gmp_inc, "-Wl,-rpath,/opt/rocm/lib/llvm/lib", "-Wl,-rpath,/opt/rocm/lib",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES] + ["-Wl," + gmp_lib]
    if verbose:
        print(" ".join(cmd))
    import time
    t0 = time.time()
    subprocess.check_call(cmd)
    LAST_ACTION = "compiled"
    # (stderr: drivers that import the package print machine-readable lines on stdout; LAST_ACTION carries the same information)
    print(f"smcpp_amd._build: compiled {os.path.relpath(LIB)} with hipcc --offload-arch=gfx950 in {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    return LIB


CY_SRC = os.path.join(HERE, "_smcpp_cy.pyx")


def cython_module_path():
    import sysconfig
    return os.path.join(HERE, "_smcpp_cy" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_cython(force: bool = False) -> str:
    """Compile the Cython binding `_smcpp_cy.pyx` (the drop-in for the reference's `_smcpp.pyx`) against the C ABI,
    in-tree next to libsmcpp_engine.so (found at run time through an $ORIGIN rpath)."""
    import sysconfig
    import numpy
    out = cython_module_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(CY_SRC),
                                                                         os.path.getmtime(os.path.join(HERE, "..", "include", "smcpp_engine.h"))):
        return out
    build(force=False)
    cpp = os.path.join(HERE, "_smcpp_cy.cpp")
    subprocess.check_call([os.environ.get("PYTHON", "python3"), "-m", "cython", "-3", "--cplus", CY_SRC, "-o", cpp])
    inc = [sysconfig.get_paths()["include"], numpy.get_include(), os.path.join(HERE, "..", "include")]
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-deprecated-declarations", "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION"]
    cmd += ["-I" + i for i in inc] + [cpp, "-o", out, "-L" + HERE, "-l:libsmcpp_engine.so", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
