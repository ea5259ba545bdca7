"""Generator of the prep-only golden G8: values produced by the compiled reference
(`oracle/_ref/libsmcpp_ref.so`, built from /root/reference by `make -C oracle ref`) for the helpers the module exposes
next to the inference managers — `PyRateFunction.R / average_coal_times / random_coal_times`, `raw_sfs` (full and
below-only) and the transition matrix — on three parameter sets.  Run here only (the reference does not travel):
    python tests/golden/make_golden_prep.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref                     # noqa: E402
from smcpp_amd import synth                # noqa: E402


def main():
    sets = []
    # S0: the constant-size model of the reference's own known-answer test (test/unit/test_bugs.py:18-33)
    sets.append((np.array([1.0]), np.array([1.0])))
    # S1: the benchmark model
    a, s = synth.model_pieces()
    sets.append((np.asarray(a), np.asarray(s)))
    # S2: a bottleneck with sizes spanning four orders of magnitude
    sets.append((np.array([5.0, 0.02, 0.5, 30.0, 1.0, 2.5]), np.array([0.002, 0.01, 0.05, 0.2, 0.7, 1.0])))
    out = {}
    seeds = np.array([1, 2, 3, 12345, 2 ** 40 + 7], dtype=np.int64)
    for si, (a, s) in enumerate(sets):
        t = np.array([0.0, 1e-3, 0.0137, 0.2, 1.08, 2.0, 7.5])
        hs = np.concatenate([[0.0], np.logspace(-2, 1, 7), [np.inf]])
        M = len(hs) - 1
        out[f"S{si}_a"], out[f"S{si}_s"], out[f"S{si}_t"], out[f"S{si}_hs"] = a, s, t, hs
        r = ref.rate(a, s, t, 0.01, 0.5, seeds)
        out[f"S{si}_R"], out[f"S{si}_random_t"], out[f"S{si}_random_R"] = r["R"], r["random_t"], r["random_R"]
        p = ref.prep(a, s, hs, rho=synth.RHO, theta=synth.THETA)
        out[f"S{si}_pi"], out[f"S{si}_T"], out[f"S{si}_avg_ct"] = p["pi"], p["T"], p["avg_ct"]
        for n in (0, 2, 10, 20):
            for iv, (t1, t2) in enumerate([(0.0, np.inf), (0.0, 0.5), (0.5, 2.0), (2.0, np.inf)]):
                raw = ref.prep(a, s, np.array([t1, t2]), rho=synth.RHO, theta=synth.THETA, n=n, raw=True)["raw_csfs"][0]
                below = ref.rate(a, s, [0.0], t1, t2, (), n=n)["below"]
                out[f"S{si}_sfs_n{n}_i{iv}"] = raw
                out[f"S{si}_below_n{n}_i{iv}"] = below
    out["seeds"] = seeds
    out["intervals"] = np.array([(0.0, np.inf), (0.0, 0.5), (0.5, 2.0), (2.0, np.inf)])
    np.savez_compressed(os.path.join(HERE, "G8_prep_only.npz"), **out)
    print("G8_prep_only written:", len(out), "arrays")


if __name__ == "__main__":
    main()
