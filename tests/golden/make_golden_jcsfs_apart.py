"""Emit golden G12: JointCSFS for one distinguished lineage per population (a1 = a2 = 1), src/jcsfs.cpp:258-367.

BUILD CONTAINER ONLY.  jcsfs.cpp itself cannot be compiled here (GSL); the golden is assembled by
``oracle/jcsfs_apart_oracle.py`` from the COMPILED reference building blocks (shiftParams / truncateParams,
OnePopConditionedSFS::compute, PiecewiseConstantRateFunction::R, modified_moran_rate_matrix) — see that file for
exactly which lines of the reference are executed and which are restated.

    python tests/golden/make_golden_jcsfs_apart.py     # writes tests/golden/G12_jcsfs_apart.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import jcsfs_apart_oracle as JA   # noqa: E402


def main():
    k = np.arange(8)
    a1 = 1.0 + 0.5 * np.sin(k); a2 = 1.5 + 0.5 * np.cos(k)
    s = np.r_[0.01, 0.01 * (1.6 ** k[1:] - 1.6 ** (k[1:] - 1))]
    cases = []
    for (n1, n2) in ((3, 2), (6, 5), (1, 4), (0, 3), (10, 10)):
        for split in (0.0, 0.005, 0.07, 0.6):
            M = 12 if n1 + n2 < 20 else 6
            hs = np.r_[0.0, np.logspace(-2, 0.8, M - 1), np.inf]
            cases.append((n1, n2, split, hs))
    out = {"a1": a1, "a2": a2, "s": s, "ncases": np.array(len(cases))}
    for i, (n1, n2, split, hs) in enumerate(cases):
        out[f"c{i}_n"] = np.array([n1, n2]); out[f"c{i}_split"] = np.array(split); out[f"c{i}_hs"] = hs
        out[f"c{i}_J"] = JA.joint_csfs_apart(n1, n2, hs, (a1, s), (a2, s), split)
    path = os.path.join(HERE, "G12_jcsfs_apart.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
