"""Emit golden G23: inputs / outputs of the reference's compiled pre-HMM data shaping `smcpp/_estimation_tools.pyx`
(SURVEY.md 8 f-2): `thin_data` (8-84), `bin_observations` (146-173), `realign` (176-209), `windowed_mutation_counts` (212-255).

BUILD CONTAINER ONLY (needs /root/reference, cython, g++).  The module as a whole cannot be built here: its first declaration is
`cdef extern from "<gsl/gsl_sf_gamma.h>"` and GSL is not in this image.  Exactly ONE function uses that declaration -
`beta_de_avg_pdf` (258-273, `gsl_sf_lnbeta`).  As oracle/Makefile leaves the two GSL translation units of the C++ out of the
compiled reference instead of stubbing GSL, this script leaves that one function and the extern block OUT of a scratch copy of
the file (nothing is written in their place, no header or library stand-in exists anywhere), compiles the remainder - the
reference's own text of the four functions above and their helpers, byte for byte - with Cython in a temporary directory, runs
it, and stores DATA only: the int32 inputs and outputs.  The scratch copy and the compiled module never enter the repository.

    python tests/golden/make_golden_estimation_tools.py        # writes tests/golden/G23_estimation_tools.npz
"""
import gzip
import importlib.util
import os
import re
import subprocess
import sys
import sysconfig
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_PYX = "/root/reference/smcpp/_estimation_tools.pyx"


def build_reference_module():
    src = open(REF_PYX).read()
    # leave out (not replace) the GSL extern block and the one function that calls it
    m = re.search(r'cdef extern from "<gsl/gsl_sf_gamma\.h>":\n(?:[ \t]+.*\n)+', src)
    assert m, "the extern block moved"
    src = src[:m.start()] + src[m.end():]
    k = src.index("def beta_de_avg_pdf(")
    tail = src[k:]
    nxt = re.search(r"\n(?=def |cdef |cpdef )", tail[1:])
    src = src[:k] + (tail[1 + nxt.end():] if nxt else "")
    assert "gsl" not in src
    for fn in ("def thin_data(", "def bin_observations(", "def realign(", "def windowed_mutation_counts(", "cdef void process_bin("):
        assert fn in src
    d = tempfile.mkdtemp(prefix="ref_et_")
    pyx = os.path.join(d, "ref_estimation_tools.pyx")
    open(pyx, "w").write(src)
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "-X", "legacy_implicit_noexcept=True", pyx, "-o", os.path.join(d, "m.c")])
    so = os.path.join(d, "ref_estimation_tools" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-Wno-deprecated-declarations", "-I" + sysconfig.get_paths()["include"],
                           "-I" + np.get_include(), os.path.join(d, "m.c"), "-o", so])
    spec = importlib.util.spec_from_file_location("ref_estimation_tools", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class C:
    """What `bin_observations` / `windowed_mutation_counts` take: an object with .data (int32 rows), .a (long per population) and
    len() = base pairs (smcpp/contig.py)."""
    def __init__(self, data, a):
        self.data = np.ascontiguousarray(data, dtype=np.int32)
        self.a = np.asarray(a, dtype=np.int64)

    def __len__(self):
        return int(self.data[:, 0].sum())


def read_smc(path):
    rows = []
    with gzip.open(path, "rt") as f:
        for line in f:
            if line.startswith("#"):
                continue
            rows.append([int(x) for x in line.split()])
    return np.array(rows, dtype=np.int32)


def main():
    ref = build_reference_module()
    from smcpp_amd import synth, vcf2smc as V
    inputs = {}
    c, _ = V.vcf2smc(os.path.join(HERE, "example.vcf.gz"), "1", ("pop1", ["msp_0", "msp_1", "msp_2"]))
    inputs["ex"] = (np.ascontiguousarray(c.data, dtype=np.int32), [2])
    inputs["chr11"] = (read_smc("/root/reference/test/bugs/11/chr11_5subjs.smc.gz")[:4000], [2])
    # two populations (7 columns): rows of the two-population generator with the spans re-expanded to base pairs and jittered
    tp = synth.synth_contig_twopop(3, 3_000_000, 6, 4).astype(np.int64)
    rng = np.random.default_rng(23)
    tp[:, 0] = tp[:, 0] * 100 - rng.integers(0, 60, len(tp))
    inputs["twopop"] = (np.ascontiguousarray(tp[:4000], dtype=np.int32), [2, 0])
    # one population with the mix `thin_data` branches on (a = 2 rows with b == nb, a = -1 rows), short spans
    r = rng.integers(0, 100, (5000, 1))
    small = np.concatenate([rng.integers(1, 40, (5000, 1)), np.where(r < 5, -1, np.where(r < 12, 2, np.where(r < 30, 1, 0))),
                            rng.integers(0, 7, (5000, 1)), np.full((5000, 1), 6)], axis=1).astype(np.int32)
    small[small[:, 1] == -1, 2:] = 0
    small[::7, 2] = small[::7, 3]                    # b == nb on some rows (the "nonseg" branch when a == 2)
    inputs["small"] = (small, [2])
    out = {}
    for name, (data, a) in inputs.items():
        out[f"{name}_in"] = data
        out[f"{name}_a"] = np.array(a, dtype=np.int64)
        # (the reference sizes its output from 2 ceil(span / thinning) rows per input row and indexes it with bounds checks: a
        # thinning far below the spans, or a large offset - which also slices ROWS off that estimate, line 19 -, raises IndexError
        # in the reference itself; such calls are outside what data_filter.py:172 ever issues and are recorded as skipped)
        big = name == "chr11"          # un-binned, spans to 1e8: small thinnings / windows give millions of rows there
        for th, off in ((400, 0), (400, 133), (1000, 0), (1000, 999), (2300, 0)) if big else ((40, 0), (400, 0), (400, 133), (1000, 0), (1000, 999), (2300, 0)):
            try:
                out[f"{name}_thin_{th}_{off}"] = ref.thin_data(data.copy(), th, off)
            except IndexError:
                print(f"   reference raises IndexError: thin_data({name}, {th}, {off})")
        for w in (500, 1000) if big else (50, 100, 1000):
            out[f"{name}_bin_{w}"] = ref.bin_observations(C(data.copy(), a), w)
            out[f"{name}_realign_{w}"] = ref.realign(data.copy(), w)
            out[f"{name}_wmc_{w}"] = np.ascontiguousarray(ref.windowed_mutation_counts(C(data.copy(), a), w))
        # the pipeline's own composition (data_filter.py:166-203: Thin, then Bin on the thinned rows)
        t = ref.thin_data(data.copy(), 400, 0)
        out[f"{name}_thin400_bin100"] = ref.bin_observations(C(t, a), 1000 if big else 100)
    # outputs on un-binned data run to millions of rows: beyond 64 KB an output is stored as its shape, the CRC-32 of its bytes
    # (int32, C order) and its first and last 500 rows - the comparison is bit-exact either way
    import zlib
    packed = {}
    for k, v in out.items():
        v = np.ascontiguousarray(v)
        if k.endswith(("_in", "_a")) or v.nbytes <= 65536:
            packed[k] = v
        else:
            assert v.dtype == np.int32
            packed[k + "__shape"] = np.array(v.shape, dtype=np.int64)
            packed[k + "__crc"] = np.array(zlib.crc32(v.tobytes()), dtype=np.int64)
            flat = v if v.shape[0] >= v.shape[1] else np.ascontiguousarray(v.T)      # (windowed_mutation_counts returns [2, windows])
            packed[k + "__head"] = flat[:500].copy()
            packed[k + "__tail"] = flat[-500:].copy()
    out = packed
    path = os.path.join(HERE, "G23_estimation_tools.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
