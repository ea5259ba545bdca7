"""Generates the golden vectors under tests/golden/ from the COMPILED REFERENCE (oracle/_ref, i.e. the reference's own
src/hmm.cpp, src/transition_bundle.cpp, src/transition.cpp, src/conditioned_sfs.cpp ... built where they lie under
/root/reference by oracle/Makefile).  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden.py

Every fixture holds inputs (pi, T, keys, E, obs ...) and the reference's outputs (loglik, Q, xisum, gamma_sums, the
posterior argmax with its top-1/top-2 margin, sub-sampled alpha_hat / gamma columns).  Fixtures are data only.
The emission table is assembled from the reference's conditioned SFS by oracle/prep_oracle.py (the reference's own
assembly lives in inference_manager.cpp, which needs GSL and cannot be built here); that assembly is pinned by the
G1 known answer of SURVEY.md Appendix E (loglik -3108.781616833272, reproduced bit-for-bit).
"""
from __future__ import annotations

import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import prep_oracle, ref  # noqa: E402
from smcpp_amd import synth  # noqa: E402


def params_for(M, n, keys_obs, a=None, s=None, hs=None, theta=synth.THETA, rho=synth.RHO, alpha=synth.ALPHA,
               pol=synth.POLARIZATION_ERROR):
    if a is None:
        a, s = synth.model_pieces()
    if hs is None:
        hs = synth.hidden_states(M)
    p = ref.prep(a, s, hs, rho, theta, n)
    ep = prep_oracle.emission_probs(keys_obs, n, p["csfs"], p["avg_ct"], theta, alpha, pol)
    keys = np.array(list(ep.keys()), dtype=np.int32)
    E = np.array(list(ep.values()))
    return dict(pi=p["pi"], T=p["T"], keys=keys, E=E, hs=hs, a=a, s=s, theta=theta, rho=rho, alpha=alpha, pol=pol,
                n=n, avg_ct=p["avg_ct"], csfs=p["csfs"])


def golden_estep(name, par, obs, alpha_stride=16, gamma_stride=16):
    r = ref.estep(par["pi"], par["T"], par["keys"], par["E"], obs, save_gamma=True, want_alpha=True)
    g = r["gamma"]                       # [M, L+1]
    M = g.shape[0]
    arg = g.argmax(axis=0).astype(np.int16)
    if M > 1:
        srt = np.sort(g, axis=0)
        margin = (srt[-1] - srt[-2]) / srt[-1]
    else:
        margin = np.ones(g.shape[1])
    gs_keys = np.array(list(r["gamma_sums"].keys()), dtype=np.int32)
    gs_vals = np.array(list(r["gamma_sums"].values()))
    out = dict(pi=par["pi"], T=par["T"], keys=par["keys"], E=par["E"], hs=par["hs"], obs=obs,
               loglik=r["loglik"], q=r["q"], xisum=r["xisum"], gs_keys=gs_keys, gs_vals=gs_vals,
               gamma0=g[:, 0].copy(), gamma_argmax=arg, gamma_margin=margin.astype(np.float32),
               gamma_sub=g[:, ::gamma_stride].copy(), gamma_stride=gamma_stride,
               alpha_sub=r["alpha_hat"][::alpha_stride].copy(), alpha_stride=alpha_stride,
               log_c=r["log_c"], a=par["a"], s=par["s"], theta=par["theta"], rho=par["rho"], alpha=par["alpha"],
               pol=par["pol"], n=par["n"])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: L={len(obs)} M={M} K={len(par['keys'])} loglik={r['loglik']!r} "
          f"min margin={margin.min():.3g} -> {os.path.getsize(path)/1024:.0f} KiB")


def g1():
    M = 16; n = 4
    hs = np.r_[0., np.logspace(-2, 1, M - 1), np.inf]
    rng = np.random.RandomState(1)
    rows = []
    for i in range(3000):
        a = rng.choice([0, 1, -1], p=[.9, .08, .02])
        if a == -1:
            rows.append([rng.randint(1, 50), -1, 0, 0]); continue
        if i % 10 == 0:
            rows.append([1, a, rng.randint(0, n + 1), n])
        else:
            rows.append([rng.randint(1, 200) if a == 0 else 1, a, 0, 0])
    obs = np.ascontiguousarray(rows, dtype=np.int32)
    par = params_for(M, n, obs[:, 1:], a=np.array([1.0, 2.0, 0.5, 1.0]), s=np.array([0.05, 0.2, 1.0, 1.0]), hs=hs,
                     theta=2.5e-2, rho=6e-3, alpha=1.0, pol=0.5)
    golden_estep("G1_M16_n4", par, obs, alpha_stride=4, gamma_stride=4)


def load_smc(path):
    rows = []
    with gzip.open(path, "rt") as f:
        for line in f:
            if line.startswith("#"):
                continue
            rows.append([int(x) for x in line.split()])
    return np.ascontiguousarray(rows, dtype=np.int32)


def main():
    if not ref.available():
        raise SystemExit("oracle/_ref/libsmcpp_ref.so missing: run `make -C oracle ref` in the build container")
    g1()
    # G3 / G4: first 2 Mbp of the synthetic contig 0 at the C2 and headline shapes; parameters cover every key of
    # the full 100 Mbp contig so the same files parameterise bench.py and the full-size property tests
    for (M, n, tag) in ((32, 10, "G3"), (64, 20, "G4")):
        full = synth.synth_contig(0, 100_000_000, n)
        par = params_for(M, n, full[:, 1:])
        np.savez_compressed(os.path.join(HERE, f"params_M{M}_n{n}.npz"), pi=par["pi"], T=par["T"], keys=par["keys"],
                            E=par["E"], hs=par["hs"], a=par["a"], s=par["s"], theta=par["theta"], rho=par["rho"],
                            alpha=par["alpha"], pol=par["pol"], n=n, rows_100mbp=len(full),
                            crc_100mbp=synth.contig_crc(full), avg_ct=par["avg_ct"], csfs=par["csfs"])
        obs = synth.synth_contig(0, 2_000_000, n)
        golden_estep(f"{tag}_M{M}_n{n}_2Mbp", par, obs)
    # G6: a single hidden state (the bootstrap manager of Analysis, analysis.py:28-57)
    obs = synth.synth_contig(1, 200_000, 4)
    par = params_for(1, 4, obs[:, 1:], hs=np.array([0.0, np.inf]))
    golden_estep("G6_M1_n4", par, obs, alpha_stride=1, gamma_stride=1)
    # G2-like: M not a multiple of 16/4, very long spans (test_inference.py:35-60 geometry)
    M = 51; n = 6
    rng = np.random.RandomState(7)
    rows = []
    for i in range(240):
        r = rng.rand()
        if r < 0.3:
            rows.append([int(rng.randint(1000, 200001)), 0, 0, 0])
        elif r < 0.4:
            rows.append([int(rng.randint(2, 400)), -1, 0, 0])
        elif r < 0.7:
            rows.append([1, int(rng.randint(0, 2)), int(rng.randint(0, n + 1)), n])
        elif r < 0.85:
            rows.append([1, 1, 0, 0])
        else:
            rows.append([int(rng.randint(2, 30)), 0, int(rng.randint(0, 3)), n - int(rng.randint(0, 3))])
    obs = np.ascontiguousarray(rows, dtype=np.int32)
    keep = np.ones(len(obs), bool)
    keep[1:] = np.any(obs[1:, 1:] != obs[:-1, 1:], axis=1)   # no equal consecutive keys (RLE'd data never has them)
    obs = np.ascontiguousarray(obs[keep])
    par = params_for(M, n, obs[:, 1:], hs=np.r_[0., np.logspace(-2, 1, M - 1), np.inf], theta=2.5e-4, rho=6.25e-5)
    golden_estep("G2_M51_n6_longspans", par, obs, alpha_stride=1, gamma_stride=1)
    # G7: the reference's own un-binned test contig (test/bugs/11/chr11_5subjs.smc.gz; data file, spans to 2e5),
    # with the missing row `smc++ posterior` prepends (commands/posterior.py:83)
    smc = load_smc("/root/reference/test/bugs/11/chr11_5subjs.smc.gz")
    obs = np.ascontiguousarray(np.vstack([[1, -1, 0, 0], smc]), dtype=np.int32)
    par = params_for(32, 8, obs[:, 1:], theta=2.5e-4, rho=6.25e-5)
    golden_estep("G7_M32_n8_chr11", par, obs, alpha_stride=1, gamma_stride=1)
    # G5-plumbing: two-population row layout (7 columns, 6-int keys).  The JointCSFS emissions need GSL-dependent
    # reference code that cannot be built here, so the emission vectors are products of two one-population vectors:
    # positive, key-dependent, and enough to pin the HMM path for keylen = 6 against hmm.cpp.
    M = 48
    obs = synth.synth_contig_twopop(2, 1_000_000, 10, 10)
    p1 = params_for(M, 10, obs[:, 1:4])
    e1 = {tuple(k): v for k, v in zip(p1["keys"].tolist(), p1["E"])}
    keys6 = np.unique(obs[:, 1:], axis=0)
    E6 = np.array([e1[tuple(k[:3])] * (0.5 + 0.5 * e1[tuple(k[:3])] ** (1 + (k[4] % 3))) if k[5] else e1[tuple(k[:3])]
                   for k in keys6.tolist()])
    par = dict(p1)
    par["keys"] = keys6.astype(np.int32)
    par["E"] = E6
    golden_estep("G5_M48_twopop_layout", par, obs)
    # bench / roofline parameters of config C5 (M=256, n=50)
    full = synth.synth_contig(0, 100_000_000, 50)
    par = params_for(256, 50, full[:, 1:])
    np.savez_compressed(os.path.join(HERE, "params_M256_n50.npz"), pi=par["pi"], T=par["T"], keys=par["keys"],
                        E=par["E"], hs=par["hs"], a=par["a"], s=par["s"], theta=par["theta"], rho=par["rho"],
                        alpha=par["alpha"], pol=par["pol"], n=50, rows_100mbp=len(full),
                        crc_100mbp=synth.contig_crc(full))
    print("params_M256_n50 written")


if __name__ == "__main__":
    main()
