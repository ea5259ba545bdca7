import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["G1_M16_n4", "G2_M51_n6_longspans", "G3_M32_n10_2Mbp", "G4_M64_n20_2Mbp", "G5_M48_twopop_layout",
                "G6_M1_n4", "G7_M32_n8_chr11", "G18_M64_n8_chr11"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.fixture
def engine_opt():
    """`engine_opt(name, value)` sets (value=None: removes) an SMCPP_* switch for the rest of the test.  The engine parses its
    option table ONCE per process (smcpp_amd/csrc/engine_options.hpp), so a change only takes effect through
    `_engine.set_option` = environment + `smcpp_reload_options`; everything is restored (and re-read) at teardown."""
    from smcpp_amd import _engine
    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = os.environ.get(name)
        _engine.set_option(name, value)
    yield set_
    for name, old in saved.items():
        _engine.set_option(name, old)
