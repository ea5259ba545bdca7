"""Golden G17: `model.final.json` as the REFERENCE writes it (`BaseAnalysis.dump`, smcpp/analysis/base.py:186-191) at a FIXED
parameter point of the example run (knots = 8, N0 = 1e4, mu = 1.25e-8, r = mu, w = 100, one population) -
no optimiser between the inputs and the file, so the comparison in tests/test_analysis.py is field by field:

  * the model: the reference's own `smcpp.model.SMCModel` on the knots its `Analysis._init_knots` places for the hidden states
    below, `y` = a fixed vector; `to_dict()` is the reference's;
  * the hidden states: the reference's `estimation_tools.balance_hidden_states(model, M + 1)` (smcpp/estimation_tools.py:170-197)
    with M = 15, its one call into the compiled binding - `PyRateFunction(model, []).R(t)` - served by the compiled reference
    (oracle/_ref: `PiecewiseConstantRateFunction::R` of src/piecewise_constant_rate_function.cpp);
  * theta / rho / alpha as `BaseAnalysis` derives them (base.py:71-107: theta = 2 N0 mu, rho = 2 N0 r, alpha = w).

BUILD CONTAINER ONLY:   make -C oracle ref && python tests/golden/make_golden_final_json.py   -> tests/golden/G17_model_final.json
The modules are imported where they lie under /root/reference (a package object named `smcpp` is created around the directory
without executing its `__init__`; `_estimation_tools` is a placeholder that raises when touched).  Data only: the JSON text.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden_model import _Raises, REF  # noqa: E402
from oracle import ref  # noqa: E402


class _RateFunction:
    """`PyRateFunction(model, hs)` (smcpp/_smcpp.pyx:370-389) for the one method balance_hidden_states calls."""

    def __init__(self, model, hs):
        self.a = np.asarray(model.stepwise_values(), dtype=float)
        self.s = np.asarray(model.s, dtype=float)

    def R(self, t):
        return float(ref.rate(self.a, self.s, [float(t)])["R"][0])


def main():
    if not hasattr(np, "VisibleDeprecationWarning"):
        np.VisibleDeprecationWarning = np.exceptions.VisibleDeprecationWarning
    pkg = types.ModuleType("smcpp")
    pkg.__path__ = [REF]
    sys.modules["smcpp"] = pkg
    et_ph = _Raises("smcpp._estimation_tools")
    sys.modules["smcpp._estimation_tools"] = et_ph
    pkg._estimation_tools = et_ph
    binding = types.ModuleType("smcpp._smcpp")
    binding.PyRateFunction = _RateFunction
    sys.modules["smcpp._smcpp"] = binding
    pkg._smcpp = binding
    model = importlib.import_module("smcpp.model")
    spline = importlib.import_module("smcpp.spline")
    ET = importlib.import_module("smcpp.estimation_tools")
    ana = importlib.import_module("smcpp.analysis.analysis")
    base = importlib.import_module("smcpp.analysis.base")

    N0, mu, w, K, M = 1e4, 1.25e-8, 100, 8, 15
    # a first model on provisional knots gives the hidden states; the knots the Analysis would place for THOSE states carry the
    # final model (analysis.py:88-126: hidden states first, then _init_knots(hs, timepoints) and the model on them)
    y = np.array([0.4, -0.3, 0.9, 0.1, -0.7, 0.25, 0.6, -0.1])
    m0 = model.SMCModel(np.geomspace(0.01, 5.0, K), N0, spline.Piecewise, "pop1")
    m0[:] = y
    hs_gen = ET.balance_hidden_states(m0, M + 1)                 # generations
    hs = hs_gen / (2.0 * N0)
    ns = types.SimpleNamespace()
    ana.Analysis._init_knots(ns, hs, None, None)
    Kf = len(ns._knots)                                         # hs[1:-1:2]: 7 knots for 15 states
    yf = np.array([0.35, -0.2, 0.8, 0.05, -0.6, 0.3, 0.5, -0.15, 0.2])[:Kf]
    m = model.SMCModel(ns._knots, N0, spline.Piecewise, "pop1")
    m[:] = yf
    self = types.SimpleNamespace(_theta=2.0 * N0 * mu, _rho=2.0 * N0 * mu, _alpha=w, model=m, hidden_states={"pop1": hs})
    path = os.path.join(HERE, "G17_model_final")
    base.BaseAnalysis.dump(self, path)
    j = json.load(open(path + ".json"))
    # the inputs of the test ride in a side file (the dump itself holds only what the reference writes)
    np.savez_compressed(os.path.join(HERE, "G17_inputs.npz"), y=y, yf=yf, knots0=np.geomspace(0.01, 5.0, K), N0=N0, mu=mu, w=w, M=M)
    print(path + ".json", os.path.getsize(path + ".json"), "bytes; keys", sorted(j), "model keys", sorted(j["model"]))


if __name__ == "__main__":
    main()
