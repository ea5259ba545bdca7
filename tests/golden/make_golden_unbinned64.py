"""Golden G18: the reference's own un-binned test contig (test/bugs/11/chr11_5subjs.smc.gz, spans to 2e5, with the missing row
`smc++ posterior` prepends) at M = 64 hidden states - the shape of `bench.py --workload posterior64` - from the COMPILED REFERENCE
(oracle/_ref), exactly as G7 (tests/golden/make_golden.py) is at M = 32:

    make -C oracle ref && python tests/golden/make_golden_unbinned64.py        -> tests/golden/G18_M64_n8_chr11.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden import golden_estep, load_smc, params_for  # noqa: E402


def main():
    smc = load_smc("/root/reference/test/bugs/11/chr11_5subjs.smc.gz")
    obs = np.ascontiguousarray(np.vstack([[1, -1, 0, 0], smc]), dtype=np.int32)
    par = params_for(64, 8, obs[:, 1:], theta=2.5e-4, rho=6.25e-5)
    golden_estep("G18_M64_n8_chr11", par, obs, alpha_stride=4, gamma_stride=4)


if __name__ == "__main__":
    main()
