"""Generator of golden G10: `Q(separate=True)` values and their gradients with respect to the
piece sizes, computed by the reference's OWN forward-mode AD (`oracle/_ref`, harness `ref_q_jac`: reference rate
function / transition / conditioned SFS on `adouble`, real `HMM::Estep` + `HMM::Q`) on the observation arrays of the
E-step goldens.  Run here only:  python tests/golden/make_golden_grad.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref                     # noqa: E402


def main():
    out = {}
    for name in ["G1_M16_n4", "G3_M32_n10_2Mbp", "G4_M64_n20_2Mbp", "G7_M32_n8_chr11"]:
        g = np.load(os.path.join(HERE, name + ".npz"))
        a = g["a"]
        q, jac, ll = ref.q_jac(g["obs"], g["keys"], int(g["n"]), a, np.eye(len(a)), g["s"], g["hs"], float(g["rho"]),
                               float(g["theta"]), float(g["alpha"]), float(g["pol"]))
        assert abs(ll - float(g["loglik"])) <= 1e-12 * abs(ll), (name, ll, float(g["loglik"]))
        np.testing.assert_allclose(q, g["q"], rtol=1e-12)
        out[name + "_q"] = q
        out[name + "_jac"] = jac
        print(name, q, np.abs(jac).max())
    np.savez_compressed(os.path.join(HERE, "G10_q_gradients.npz"), **out)


if __name__ == "__main__":
    main()
