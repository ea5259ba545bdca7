"""Emit golden G11: row-for-row inputs / outputs of the reference's pre-HMM data shaping.

BUILD CONTAINER ONLY (needs /root/reference).  The reference's pure-Python shaping code is imported where it lies
(`smcpp/contig.py`, `smcpp/estimation_tools.py`, `smcpp/data_filter.py`) under a scratch package name, without the
package `__init__` (which needs the compiled binding).  Two of their imports cannot be satisfied here and are bound
to placeholders whose every attribute RAISES when touched, so nothing in the fixture can come from them:

  * `smcpp/_estimation_tools.pyx` (thin_data, bin_observations, windowed_mutation_counts, realign) is Cython with
    `cdef extern from "<gsl/gsl_sf_gamma.h>"`; GSL is not in this image, so that module is unbuildable here and its
    four functions stay pinned by the reference's own known answers (test/unit/test_bugs.py:35-47) and the survey's
    recorded pipeline shapes only (tests/test_data.py says so);
  * `smcpp/model.py` does `import smcpp` (the compiled package).

What is captured (all executed by the reference's code on real data):
  compress_repeated_obs           estimation_tools.py:51-60
  decompress_polymorphic_spans    estimation_tools.py:63-85
  recode_nonseg                   estimation_tools.py:88-114
  break_long_spans                estimation_tools.py:117-167
  _load_data_helper (.smc reader) estimation_tools.py:236-267
  Validate / Watterson / RecodeMonomorphic / DropSmallContigs / DropUninformativeContigs   data_filter.py
on (a) the contig this repository's VCF converter makes of the reference's example VCF (an input: the reference's
converter needs pysam) and (b) the reference's own test data test/bugs/11/*.smc.gz.

    python tests/golden/make_golden_pipeline.py        # writes tests/golden/G11_pipeline.npz
"""
import gzip
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/smcpp"


class _Raises(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def f(*a, **k):
            raise NotImplementedError(f"{self.__name__}.{name} is not available in this container")
        return f


def load_reference():
    pkg = types.ModuleType("smcpp_ref")
    pkg.__path__ = [REF]
    sys.modules["smcpp_ref"] = pkg
    for ph in ("_estimation_tools", "model"):
        m = _Raises("smcpp_ref." + ph)
        sys.modules["smcpp_ref." + ph] = m
        setattr(pkg, ph, m)
    mods = {}
    for name in ("contig", "defaults", "util", "estimation_tools", "data_filter"):
        spec = importlib.util.spec_from_file_location("smcpp_ref." + name, os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["smcpp_ref." + name] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
        mods[name] = m
    return mods


def main():
    ref = load_reference()
    ET, DF, Contig = ref["estimation_tools"], ref["data_filter"], ref["contig"].Contig
    from smcpp_amd import vcf2smc as V
    out = {}

    # ---- (a) the example-derived contig ----
    c, _ = V.vcf2smc(os.path.join(HERE, "example.vcf.gz"), "1", ("pop1", ["msp_0", "msp_1", "msp_2"]))
    raw = np.ascontiguousarray(c.data, dtype=np.int32)
    out["ex_raw"] = raw
    out["ex_n"] = np.array(c.n); out["ex_a"] = np.array(c.a)
    out["ex_compress"] = ET.compress_repeated_obs(raw.copy())
    dec = ET.decompress_polymorphic_spans(out["ex_compress"].copy())
    out["ex_decompress"] = dec
    for cutoff in (100000, 5000, 2000):
        pieces = ET.break_long_spans(Contig(pid=("pop1",), data=out["ex_compress"].copy(), n=c.n, a=c.a, fn="ex"), cutoff)
        out[f"ex_break_{cutoff}_n"] = np.array(len(pieces))
        for i, p in enumerate(pieces):
            out[f"ex_break_{cutoff}_{i}"] = p.data
    for cutoff in (None, 3000):
        cc = Contig(pid=("pop1",), data=out["ex_compress"].copy(), n=c.n, a=c.a, fn="ex")
        out[f"ex_recode_nonseg_{cutoff}"] = ET.recode_nonseg(cc, cutoff).data
    w = DF.Watterson()
    w.run([Contig(pid=("pop1",), data=raw.copy(), n=c.n, a=c.a, fn="ex")])
    out["ex_watterson"] = np.array(w.theta_hat)
    out["ex_validate"] = DF.Validate().run(Contig(pid=("pop1",), data=raw.copy(), n=c.n, a=c.a, fn="ex")).data
    out["ex_recode_mono"] = DF.RecodeMonomorphic().run([Contig(pid=("pop1",), data=raw.copy(), n=c.n, a=c.a, fn="ex")])[0].data

    # ---- (b) the reference's own test data: reader + the same chain ----
    bug = "/root/reference/test/bugs/11"
    files = sorted(f for f in os.listdir(bug) if f.endswith(".smc.gz"))
    out["bug11_files"] = np.array(files)
    for fi, f in enumerate(files):
        contig = ET._load_data_helper(os.path.join(bug, f))
        d = np.ascontiguousarray(contig.data, dtype=np.int32)
        # the text itself is the input of this repository's reader: keep the first 4 000 data lines as the fixture
        with gzip.open(os.path.join(bug, f), "rt") as fh:
            lines = fh.read().splitlines()
        head = [ln for ln in lines if ln.startswith("#")]
        body = [ln for ln in lines if not ln.startswith("#")][:4000]
        out[f"bug11_{fi}_text"] = np.array("\n".join(head + body) + "\n")
        out[f"bug11_{fi}_rows"] = np.array(d.shape[0])
        out[f"bug11_{fi}_pid"] = np.array(list(contig.pid))
        out[f"bug11_{fi}_n"] = np.array(contig.n); out[f"bug11_{fi}_a"] = np.array(contig.a)
        d = d[:4000].copy()
        out[f"bug11_{fi}_data"] = d
        out[f"bug11_{fi}_compress"] = ET.compress_repeated_obs(d.copy())
        cc = Contig(pid=contig.pid, data=out[f"bug11_{fi}_compress"].copy(), n=contig.n, a=contig.a, fn=f)
        pieces = ET.break_long_spans(cc, 20000)
        out[f"bug11_{fi}_break_n"] = np.array(len(pieces))
        for i, p in enumerate(pieces):
            out[f"bug11_{fi}_break_{i}"] = p.data
        cc = Contig(pid=contig.pid, data=out[f"bug11_{fi}_compress"].copy(), n=contig.n, a=contig.a, fn=f)
        out[f"bug11_{fi}_recode_nonseg"] = ET.recode_nonseg(cc, 10000).data
        w = DF.Watterson()
        w.run([Contig(pid=contig.pid, data=d.copy(), n=contig.n, a=contig.a, fn=f)])
        out[f"bug11_{fi}_watterson"] = np.array(w.theta_hat)
        out[f"bug11_{fi}_recode_mono"] = DF.RecodeMonomorphic().run(
            [Contig(pid=contig.pid, data=d.copy(), n=contig.n, a=contig.a, fn=f)])[0].data
    # ---- (c) Validate on malformed rows: `span <= 0 | A | B | C` parses as `span <= (0 | A | B | C)` (data_filter.py:146-150),
    # so a row with span > 1 passes whatever its counts are; which of these one-violation contigs the reference refuses ----
    cases = np.array([[1, 3, 0, 0], [5, 3, 0, 0], [1, 0, 3, 2], [7, 0, 3, 2], [1, 0, 0, 9], [2, 0, 0, 9], [0, 0, 0, 0], [4, 1, 2, 4]],
                     dtype=np.int32)
    raised = []
    for row in cases:
        d = np.array([[3, 0, 0, 0], row, [2, 1, 1, 4]], dtype=np.int32)
        try:
            DF.Validate().run(Contig(pid=("pop1",), data=np.ascontiguousarray(d), n=np.array([4]), a=np.array([2]), fn="case"))
            raised.append(False)
        except RuntimeError:
            raised.append(True)
    out["validate_cases"] = cases
    out["validate_raised"] = np.array(raised)

    # known answers of the reference's own unit tests for the Cython functions it cannot run here (test_bugs.py:35-47)
    out["kat_compress_in"] = np.array([[1, 0, 0, 0], [2, 0, 0, 0]], dtype=np.int32)
    out["kat_compress_out"] = ET.compress_repeated_obs(np.array([[1, 0, 0, 0], [2, 0, 0, 0]]))
    path = os.path.join(HERE, "G11_pipeline.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
