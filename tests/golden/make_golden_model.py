"""Golden G15: the reference's own pure-Python model classes (smcpp/model.py, smcpp/spline/piecewise.py, smcpp/defaults.py) and
the knot placement / regularisation scaling of its Analysis (smcpp/analysis/analysis.py:104-126), imported where they lie under
/root/reference (build container only):

    python tests/golden/make_golden_model.py

`smcpp/__init__.py` needs the compiled binding, so a package object named `smcpp` is created around the reference's directory
WITHOUT executing its `__init__`; the compiled modules it would import (`_smcpp`, `_estimation_tools`) are placeholders that
raise when touched.  Only data is written: inputs and the reference's outputs (SMCModel.s / stepwise_values / __call__ /
regularizer / to_dict, randomize under a fixed seed, Analysis._init_knots, the regularisation penalty)."""
from __future__ import annotations

import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/smcpp"


class _Raises(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def f(*a, **k):
            raise NotImplementedError(f"{self.__name__}.{name} is not available in this container")
        return f


def load_reference():
    if not hasattr(np, "VisibleDeprecationWarning"):
        np.VisibleDeprecationWarning = np.exceptions.VisibleDeprecationWarning
    pkg = types.ModuleType("smcpp")
    pkg.__path__ = [REF]
    sys.modules["smcpp"] = pkg
    for ph in ("_smcpp", "_estimation_tools"):
        m = _Raises("smcpp." + ph)
        sys.modules["smcpp." + ph] = m
        setattr(pkg, ph, m)
    model = importlib.import_module("smcpp.model")
    spline = importlib.import_module("smcpp.spline")
    return model, spline


def main():
    model, spline = load_reference()
    out = {}
    rng = np.random.default_rng(5)
    cases = []
    for ci, K in enumerate([4, 8, 15]):
        knots = np.sort(0.002 * 50.0 ** rng.random(K))
        y = rng.normal(0.0, 1.2, size=K)
        if ci == 2:
            y[0] = 9.0; y[-1] = -9.0          # clipped by minimum / maximum_population_size
        m = model.SMCModel(knots, 1e4, spline.Piecewise, "pop1")
        m[:] = y
        pts = np.r_[knots[0] / 3, knots, 0.5 * (knots[1:] + knots[:-1]), knots[-1] * 4]
        out[f"c{ci}_knots"] = knots; out[f"c{ci}_y"] = y
        out[f"c{ci}_s"] = np.asarray(m.s, dtype=float)
        out[f"c{ci}_stepwise"] = np.asarray(m.stepwise_values(), dtype=float)
        out[f"c{ci}_points"] = pts
        out[f"c{ci}_values"] = np.asarray(m(pts), dtype=float)
        out[f"c{ci}_regularizer"] = np.array(float(m.regularizer()))
        d = m.to_dict()
        out[f"c{ci}_dict"] = np.array(json.dumps(d))
        m2 = model.SMCModel.from_dict(json.loads(json.dumps(d)))
        assert np.array_equal(np.asarray(m2.stepwise_values(), dtype=float), out[f"c{ci}_stepwise"])
        np.random.seed(3)
        m.randomize()
        out[f"c{ci}_randomized"] = np.asarray(m[:], dtype=float)
        cases.append(K)
    out["n_cases"] = np.array(len(cases))
    # Analysis._init_knots / _init_regularization on a stand-in for `self` (they only touch self._knots / self._args / self.Q)
    try:
        ana = importlib.import_module("smcpp.analysis.analysis")
        Analysis = ana.Analysis
    except Exception as e:  # noqa: BLE001
        raise SystemExit(f"cannot import the reference's Analysis: {e!r}")
    hs_sets = [np.r_[0.0, np.sort(0.01 * 30.0 ** rng.random(15)), np.inf], np.r_[0.0, np.geomspace(0.02, 5.0, 9), np.inf]]
    k = 0
    for hs in hs_sets:
        for t1, tK in [(None, None), (hs[1] / 20, None), (hs[1] / 7, hs[-2] * 3), (None, hs[-2] / 2)]:
            ns = types.SimpleNamespace()
            Analysis._init_knots(ns, hs, t1, tK)
            out[f"k{k}_hs"] = hs; out[f"k{k}_t"] = np.array([np.nan if t1 is None else t1, np.nan if tK is None else tK])
            out[f"k{k}_knots"] = np.asarray(ns._knots, dtype=float)
            k += 1
    out["n_knot_cases"] = np.array(k)
    pens = []
    for q, rp, lam in [(-1234.5, 6, None), (-0.02, 3, None), (55.0, 5.5, None), (-10.0, 6, 0.25)]:
        ns = types.SimpleNamespace(_args=types.SimpleNamespace(lambda_=lam), Q=lambda q=q: q)
        Analysis._init_regularization(ns, types.SimpleNamespace(lambda_=lam, regularization_penalty=rp))
        pens.append([q, rp, np.nan if lam is None else lam, ns._penalty])
    out["penalty_cases"] = np.array(pens)
    path = os.path.join(HERE, "G15_model.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
