"""Generator of golden G9 (joint CSFS, distinguished pair in population 1).

The reference has two implementations of the joint conditioned SFS: `src/jcsfs.cpp` (needs GSL headers, which this
image lacks, so it cannot be built) and the pure-Python original `smcpp/jcsfs.py`, of which the C++ file is a
line-by-line translation (include/jcsfs.h:11-12).  This script imports the Python original from /root/reference —
without running the package's `__init__` (it pulls in the compiled binding) — and serves the three calls it makes into
`smcpp._smcpp` from the compiled reference C++ (`oracle/_ref`, built by `make -C oracle ref`):
    raw_sfs(model, n, t1, t2, below_only)   -> OnePopConditionedSFS::compute / compute_below
    PyRateFunction(model, []).R(t)          -> PiecewiseConstantRateFunction::R
    PyRateFunction.random_coal_times        -> random_time(1., t1, t2, gen) on ONE default-seeded std::mt19937, i.e.
                                               the draw sequence of jcsfs.cpp:120-127 (the Python original seeds from
                                               numpy instead; using the C++ sequence makes the golden deterministic and
                                               equal to what the C++ translation computes).
Run here only:  python tests/golden/make_golden_jcsfs.py
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref                     # noqa: E402


def _as(model):
    return (np.array([float(x) for x in model.stepwise_values()]), np.array(model.s, dtype=float))


def load_reference_jcsfs():
    pkg = types.ModuleType("smcpp")
    pkg.__path__ = ["/root/reference/smcpp"]
    sys.modules["smcpp"] = pkg
    shim = types.ModuleType("smcpp._smcpp")

    def raw_sfs(model, n, t1, t2, below_only=False):
        a, s = _as(model)
        if below_only:
            return ref.rate(a, s, [0.0], t1, t2, (), n=n)["below"]
        return ref.prep(a, s, np.array([t1, t2]), rho=1e-3, theta=1e-3, n=n, raw=True)["raw_csfs"][0]

    class PyRateFunction:
        def __init__(self, model, hs):
            self._a, self._s = _as(model)

        def R(self, t):
            return float(ref.rate(self._a, self._s, [t])["R"][0])

        def random_coal_times(self, t1, t2, K):
            t, R = ref.random_times_shared(self._a, self._s, t1, t2, K)
            return list(zip(t, R))

    shim.raw_sfs = raw_sfs
    shim.PyRateFunction = PyRateFunction
    sys.modules["smcpp._smcpp"] = shim
    pkg._smcpp = shim

    # jcsfs.py predates the current smcpp/model.py: it builds its helper models as PiecewiseModel(s, a, dlist)
    # (jcsfs.py:262,277 pass the piece LENGTHS first), while model.py:60 now takes (a, s, N0).  Loading the current
    # model.py would silently swap sizes and lengths, so the three attributes jcsfs.py uses are provided here with the
    # argument order it was written for.
    mshim = types.ModuleType("smcpp.model")

    class PiecewiseModel:
        def __init__(self, s, a, dlist=None):
            self.s = np.array(s, dtype=float)
            self.a = np.array(a, dtype=float)
            self.dlist = []

        def stepwise_values(self):
            return self.a

    mshim.PiecewiseModel = PiecewiseModel
    sys.modules["smcpp.model"] = mshim
    pkg.model = mshim
    return importlib.import_module("smcpp.jcsfs"), mshim


def main():
    jmod, mmod = load_reference_jcsfs()
    out = {}
    cases = [
        # name, n1, n2, hidden states, K, model1 (a, s), model2 (a, s), splits
        ("A", 5, 2, [0.0, 0.5, 1.0, np.inf], 10, ([1.0, 4.0], [0.5, 1.0]), ([2.0, 4.0, 2.0], [0.1, 0.2, 0.3]),
         [0.1, 0.25, 0.5, 0.75, 1.5]),
        ("B", 7, 5, [0.0, 0.5, 1.0, 2.0, np.inf], 25, ([1.0, 4.0], [0.5, 1.0]), ([2.0, 4.0, 2.0], [0.1, 0.2, 0.3]),
         [0.5, 1.0, 2.0]),
        ("C", 10, 10, [0.0, 0.02, 0.1, 0.4, 1.2, 3.0, np.inf], 10,
         ([3.0, 0.3, 1.5, 0.8, 2.0], [0.01, 0.05, 0.2, 0.5, 1.0]), ([0.5, 6.0, 1.0], [0.03, 0.3, 1.0]), [0.07, 0.4, 0.9]),
        ("D", 4, 1, [0.0, 1.0, np.inf], 10, ([1.0], [1.0]), ([2.5], [1.0]), [0.3]),
        ("E", 3, 0, [0.0, 0.7, np.inf], 10, ([1.0, 0.5], [0.2, 1.0]), ([2.5], [1.0]), [0.3]),
    ]
    for name, n1, n2, hs, K, (a1, s1), (a2, s2), splits in cases:
        m1 = mmod.PiecewiseModel(np.array(s1), np.array(a1))
        m2 = mmod.PiecewiseModel(np.array(s2), np.array(a2))
        j = jmod.JointCSFS(n1, n2, 2, 0, hs, K)
        res = np.array([j.compute(m1, m2, sp).astype(float).copy() for sp in splits])
        out[f"{name}_n"] = np.array([n1, n2, K])
        out[f"{name}_hs"] = np.array(hs)
        out[f"{name}_a1"], out[f"{name}_s1"] = np.array(a1), np.array(s1)
        out[f"{name}_a2"], out[f"{name}_s2"] = np.array(a2), np.array(s2)
        out[f"{name}_splits"] = np.array(splits)
        out[f"{name}_J"] = res
        print(name, res.shape, float(res.sum()))
    np.savez_compressed(os.path.join(HERE, "G9_jcsfs_together.npz"), **out)


if __name__ == "__main__":
    main()
