"""Emit golden G13: the parameters of config C4 at its real shape (two populations, M = 48,
n1 = n2 = 10, a = (2, 0), split 0.5) from the REFERENCE: the joint CSFS by the reference's Python original
`smcpp/jcsfs.py` backed by the compiled reference C++ (the machinery of make_golden_jcsfs.py), pi / transition /
average coalescence times by the compiled reference (`ref_prep`), the 6-int-key emission table assembled from them by
the literal restatement of the reference's templates in oracle/prep_oracle.py (inference_manager.cpp needs GSL).

BUILD CONTAINER ONLY:  python tests/golden/make_golden_c4.py   -> tests/golden/G13_c4_params.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden_jcsfs import load_reference_jcsfs   # noqa: E402
from oracle import prep_oracle, ref                  # noqa: E402
from smcpp_amd import synth                           # noqa: E402


def main():
    M, n1, n2 = 48, 10, 10
    hs = synth.hidden_states(M)
    a1, s1 = synth.model_pieces()
    a2, s2 = 1.5 + 0.5 * np.cos(np.arange(8)), s1[:8].copy()
    split, theta, rho, alpha, pol = 0.5, synth.THETA, synth.RHO, synth.ALPHA, 0.5
    rows = synth.synth_contig_twopop(0, 100_000_000, n1, n2)
    keys = np.unique(rows[:, 1:], axis=0).astype(np.int32)
    jmod, mmod = load_reference_jcsfs()
    j = jmod.JointCSFS(n1, n2, 2, 0, list(hs), 10)
    J = np.array(j.compute(mmod.PiecewiseModel(np.array(s1), np.array(a1)), mmod.PiecewiseModel(np.array(s2), np.array(a2)),
                           split), dtype=float)
    # JointCSFS::compute epilogue (src/jcsfs.cpp:228-243), which the Python original leaves to its caller
    J = np.where(J > 1e-20, J, 1e-20)
    J[:, 0, 0, 0, 0] = 0.0
    J[:, 2, n1, 0, n2] = 0.0
    p = ref.prep(a1, s1, hs, rho, theta, n=-1)        # distinguished model = population 1 (both lineages there)
    tens = prep_oracle.incorporate_theta(J, theta)
    ep = prep_oracle.emission_probs_npop([tuple(k) for k in keys.tolist()], (n1, n2), (2, 0), tens, p["avg_ct"], theta,
                                         alpha, pol)
    E = np.array([ep[tuple(k)] for k in keys.tolist()])
    path = os.path.join(HERE, "G13_c4_params.npz")
    np.savez_compressed(path, hs=hs, a1=a1, s1=s1, a2=a2, s2=s2, split=split, theta=theta, rho=rho, alpha=alpha, pol=pol,
                        keys=keys, pi=p["pi"], T=p["T"], avg_ct=p["avg_ct"], J=J, E=E, n=np.array([n1, n2]))
    print(path, os.path.getsize(path), "bytes; keys", len(keys), "J", J.shape)


if __name__ == "__main__":
    main()
