"""f-2 / f-3 helpers: data format, data shaping identities (test/unit/test_bugs.py:35-47 intent), rate function known
answers (test_bugs.py:18-33) and balanced hidden states."""
import gzip
import json
import os

import numpy as np
import pytest

from conftest import load_golden


def test_rate_function_known_answers():
    from smcpp_amd import _engine
    # constant size: R(t) = t / a
    R = _engine.host_rate_function([1.0], [1.0], [0.0, 0.5, 2.0])
    np.testing.assert_allclose(R, [0.0, 0.5, 2.0], rtol=1e-14)
    R = _engine.host_rate_function([2.0, 0.5], [1.0, 1.0], [0.5, 1.0, 3.0])
    np.testing.assert_allclose(R, [0.25, 0.5, 0.5 + 4.0], rtol=1e-14)
    # E[T | t1 <= T < t2] for Exp(1)
    t1, t2 = 0.3, 1.7
    _, ct = _engine.host_rate_function([1.0], [1.0], [0.0], hs=[0.0, t1, t2, np.inf])
    ex = lambda lo, hi: ((lo + 1) * np.exp(-lo) - (hi + 1) * np.exp(-hi)) / (np.exp(-lo) - np.exp(-hi))
    np.testing.assert_allclose(ct, [ex(0, t1), ex(t1, t2), t2 + 1.0], rtol=1e-12)


def test_balance_hidden_states_constant_size():
    from smcpp_amd.model import PiecewiseModel
    from smcpp_amd.posterior import balance_hidden_states
    hs = balance_hidden_states(PiecewiseModel([1.0], [1.0]), 8)
    expect = np.r_[-np.log(1 - np.arange(8) / 8.0), np.inf]
    np.testing.assert_allclose(hs, expect, rtol=1e-9, atol=1e-12)


def test_compress_and_break(tmp_path):
    from smcpp_amd import data
    d = np.array([[3, 0, 0, 0], [2, 0, 0, 0], [1, 1, 2, 4], [1, 1, 2, 4], [500, -1, 0, 0], [4, 0, 0, 0]], dtype=np.int32)
    c = data.compress_repeated_obs(d)
    assert c.tolist() == [[5, 0, 0, 0], [2, 1, 2, 4], [500, -1, 0, 0], [4, 0, 0, 0]]
    assert c[:, 0].sum() == d[:, 0].sum()
    pieces = data.break_long_spans(data.Contig(data=c, n=[4], a=[2]), 100)
    assert len(pieces) == 2
    assert pieces[0].data.tolist() == [[1, -1, 0, 0], [5, 0, 0, 0], [2, 1, 2, 4]]
    assert pieces[1].data.tolist() == [[1, -1, 0, 0], [4, 0, 0, 0]]
    # .smc round trip incl. the column swap for a = (0, 2)
    fn = str(tmp_path / "x.smc.gz")
    hdr = {"pids": ["A", "B"], "dist": [[], [["s", 0], ["s", 1]]], "undist": [[["u", 0]], [["v", 0], ["v", 1]]], "version": "t"}
    with gzip.open(fn, "wt") as f:
        f.write("# SMC++ " + json.dumps(hdr) + "\n")
        f.write("10 0 0 1 0 0 2\n1 0 1 1 1 0 2\n")
    ct = data.load_smc(fn)
    assert ct.pid == ("B", "A") and ct.a == [2, 0] and ct.n == [2, 1]
    assert ct.data.tolist() == [[10, 0, 0, 2, 0, 0, 1], [1, 1, 0, 2, 0, 1, 1]]


def test_thinning_and_binning():
    from smcpp_amd import data
    d = np.array([[7, 0, 1, 5], [3, 1, 0, 5], [10, -1, 0, 0], [2, 2, 5, 5]], dtype=np.int32)
    t = data.thin_data(d, 4)
    assert t[:, 0].sum() == d[:, 0].sum()                      # positions are conserved (asserted in the reference too)
    # positions 3, 7 (0-based) close a window inside the first two rows and keep their SFS; 11, 15, 19 are missing
    assert t.tolist()[:5] == [[3, 0, 0, 0], [1, 0, 1, 5], [3, 0, 0, 0], [1, 1, 0, 5], [2, 1, 0, 0]]
    assert t[-1].tolist() == [2, 0, 0, 0]                      # a = 2 row recoded as non-segregating
    b = data.bin_observations(np.array([[3, 0, 0, 0], [1, 1, 0, 0], [4, 0, 0, 0], [1, 0, 2, 5], [6, -1, 0, 0]],
                                       dtype=np.int32), 5, [2])
    assert b[:, 0].tolist() == [1, 1, 1]
    assert b[0, 1:].tolist() == [1, 0, 0]                      # only the pair observed: the segregating row wins
    assert b[1, 1:].tolist() == [0, 2, 5]                      # the row with undistinguished samples wins
    assert b[2, 1:].tolist() == [-1, 0, 0]


def test_watterson_recode_and_windowed_counts():
    from smcpp_amd import data as D
    rows = np.array([[5, 0, 0, 4], [1, 1, 2, 4], [3, -1, 0, 0], [2, 0, 0, 4], [1, 2, 4, 4], [4, 1, 0, 0]], dtype=np.int32)
    c = D.Contig(rows.copy(), ("p",), [4], [2])
    # segregating rows: (1,2,4) span 1, (2,4,4) span 1, (1,0,0) span 4 -> 6; sample size as the reference counts
    # it: nb + 1 per population whose distinguished genotype is not missing
    seg = 1 + 1 + 4
    ss = np.array([5, 5, 0, 5, 5, 1.0]); sp = np.array([5, 1, 3, 2, 1, 4.0]); nm = ss > 0
    want = seg / (sp[nm] * (np.log(ss[nm]) + 0.5 / ss[nm] + 0.57721)).sum()
    assert D.watterson_theta([c]) == pytest.approx(want, rel=1e-14)
    D.recode_monomorphic(c)
    assert c.data[4].tolist() == [1, 0, 0, 4]                 # (a, b, nb) = (2, 4, 4): everything derived
    assert c.data[1].tolist() == [1, 1, 2, 4]
    big = D.Contig(np.array([[60000, 0, 0, 0], [7, 0, 0, 2], [10, 1, 0, 0]], dtype=np.int32), ("p",), [2], [2])
    D.recode_nonseg(big, None)
    assert big.data[0].tolist() == [60000, 0, 0, 0]
    D.recode_nonseg(big, 50000)
    assert big.data[0].tolist() == [60000, -1, 0, 0] and big.data[1].tolist() == [7, 0, 0, 2]
    # windows of 4 bp over 5 + 1 + 3(missing) + 2 + 1 + 4 = 16 bp; a in {1} is heterozygous
    c2 = D.Contig(rows.copy(), ("p",), [4], [2])
    wc = D.windowed_mutation_counts(c2, 4)
    assert wc.shape == (2, 16 // 4 + 1)
    # positions: 0-4 hom, 5 het, 6-8 missing, 9-10 hom, 11 (a=2, even) hom, 12-15 het
    assert wc[0].tolist() == [4, 2, 3, 4, 0] and wc[1].tolist() == [0, 1, 0, 4, 0]


def _random_rows(rng, L, n=6, npop=1):
    rows = []
    for _ in range(L):
        r = [int(rng.integers(1, 40))]
        for _p in range(npop):
            if rng.random() < 0.1:
                r += [-1, 0, 0]
            else:
                nb = int(rng.choice([0, 0, 0, n]))
                r += [int(rng.integers(0, 3)), int(rng.integers(0, nb + 1)), nb]
        rows.append(r)
    return np.array(rows, dtype=np.int32)


@pytest.mark.parametrize("seed", range(8))
def test_shaping_invariants_on_random_contigs(seed):
    """Properties the reference's pipeline relies on (`estimation_tools.py:51-60,117-167`, `_estimation_tools.pyx:8-173`):
    spans are conserved, compression is idempotent and never leaves equal neighbours, thinning keeps the full
    observation exactly once per window, binning emits one span-1 row per window."""
    from smcpp_amd import data as D
    rng = np.random.default_rng(seed)
    npop = 1 + seed % 2
    raw = _random_rows(rng, 300, npop=npop)
    total = int(raw[:, 0].sum())
    c = D.compress_repeated_obs(raw)
    assert int(c[:, 0].sum()) == total
    assert np.all(np.any(c[1:, 1:] != c[:-1, 1:], axis=1))
    np.testing.assert_array_equal(D.compress_repeated_obs(c), c)
    # expanding both to one row per base gives the same sequence
    np.testing.assert_array_equal(np.repeat(raw[:, 1:], raw[:, 0], axis=0), np.repeat(c[:, 1:], c[:, 0], axis=0))
    thinning = int(rng.integers(5, 60))
    t = D.thin_data(raw, thinning)
    assert int(t[:, 0].sum()) == total
    per_base_raw = np.repeat(raw[:, 1:], raw[:, 0], axis=0)
    per_base_thin = np.repeat(t[:, 1:], t[:, 0], axis=0)
    keep = np.arange(total) % thinning == thinning - 1          # last position of every window
    sa2 = per_base_raw[:, 0::3].sum(axis=1) == 2
    # kept positions carry the full observation (unless the distinguished counts sum to 2: recoded to zeros)
    np.testing.assert_array_equal(per_base_thin[keep & ~sa2], per_base_raw[keep & ~sa2])
    assert np.all(per_base_thin[keep & sa2] == 0)
    # every other position keeps the distinguished counts only
    other = ~keep
    assert np.all(per_base_thin[other][:, 1::3] == 0) and np.all(per_base_thin[other][:, 2::3] == 0)
    np.testing.assert_array_equal(per_base_thin[other & ~sa2][:, 0::3], per_base_raw[other & ~sa2][:, 0::3])
    w = int(rng.integers(3, 25))
    b = D.bin_observations(raw, w, [2] + [0] * (npop - 1))
    assert np.all(b[:, 0] == 1)
    assert len(b) in (total // w, total // w + 1, -(-total // w))
    # cutting at long missing runs: the pieces put back together (minus the leading missing row of each piece and the
    # runs that were cut out) give the original rows
    raw2 = raw.copy()
    miss = np.zeros(raw.shape[1], dtype=np.int32); miss[0] = 5000; miss[1::3] = -1
    raw2 = np.insert(raw2, [50, 180], miss, axis=0)
    pieces = D.break_long_spans(D.Contig(raw2, tuple("p%d" % i for i in range(npop)), [6] * npop, [2] + [0] * (npop - 1)), 1000)
    assert len(pieces) == 3
    back = np.concatenate([p.data[1:] for p in pieces])
    np.testing.assert_array_equal(back, raw)
    for p in pieces:
        assert p.data[0, 0] == 1 and np.all(p.data[0, 1::3] == -1)


def test_vcf2smc_on_the_reference_example(tmp_path):
    """Config C1: the reference's `example/example.vcf.gz` (1 Mbp, 5 diploids; kept as a data fixture) through the VCF
    reader.  The un-binned contig for `pop1:msp_0,msp_1,msp_2` must have the shape the reference pipeline reported for it
    (measured with the reference end to end: 1 850 rows including the missing row `estimate` prepends,
    22 distinct keys, longest span 13 599)."""
    from smcpp_amd import data as D, vcf2smc as V
    vcf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example.vcf.gz")
    c, hdr = V.vcf2smc(vcf, "1", ("pop1", ["msp_0", "msp_1", "msp_2"]))
    d = c.data
    assert d.shape == (1849, 4) and int(d[:, 0].sum()) == 1_000_000
    assert len(np.unique(d[:, 1:], axis=0)) == 22 and int(d[:, 0].max()) == 13599
    assert c.n == [4] and c.a == [2]
    assert hdr["dist"] == [[["msp_0", 0], ["msp_0", 1]]] and len(hdr["undist"][0]) == 4
    assert np.all(np.any(d[1:, 1:] != d[:-1, 1:], axis=1))                 # merged like RepeatingWriter
    # first record: POS 1885, genotypes 0|0 1|1 0|1 -> a = 0, b = 3 of 4 ... cross-check against a direct parse
    lines = [l.split("\t") for l in gzip.open(vcf, "rt").read().splitlines() if not l.startswith("#")]
    pos = 0
    want = []
    for f in lines:
        p = int(f[1])
        g = [f[9 + i].split("|") for i in range(3)]
        a = -1 if "." in g[0] else int(g[0][0] != "0") + int(g[0][1] != "0")
        und = [x for x in g[1] + g[2] if x != "."]                    # missing undistinguished alleles are not counted
        b, nb = sum(int(x != "0") for x in und), len(und)
        if a == 2 and b == nb:
            a = b = 0
        if p - pos - 1 >= 1:
            want.append([p - pos - 1, 0, 0, 4])
        want.append([1, a, b, nb])
        pos = p
    want.append([1_000_000 - pos, 0, 0, 4])
    np.testing.assert_array_equal(D.compress_repeated_obs(np.array(want, dtype=np.int32)), d)
    # text round trip through the on-disk format
    out = str(tmp_path / "ex.smc.gz")
    V.write_smc(out, c, hdr)
    back = D.load_smc(out)
    np.testing.assert_array_equal(back.data, d)
    assert list(back.pid) == ["pop1"] and back.n == [4] and back.a == [2]
    # two populations: distinguished pair in pop1, pop2 undistinguished only
    c2, h2 = V.vcf2smc(vcf, "1", ("p1", ["msp_0", "msp_1"]), ("p2", ["msp_2", "msp_3"]))
    assert c2.data.shape[1] == 7 and int(c2.data[:, 0].sum()) == 1_000_000 and c2.n == [2, 4] and c2.a == [2, 0]
    # a missing cutoff turns long gaps into missing rows
    c3, _ = V.vcf2smc(vcf, "1", ("pop1", ["msp_0", "msp_1", "msp_2"]), missing_cutoff=5000)
    assert np.any((c3.data[:, 0] > 5000) & (c3.data[:, 1] == -1)) and int(c3.data[:, 0].sum()) == 1_000_000


def test_reference_pipeline_shape_on_the_example():
    """The reference's `estimate` pipeline on its example contig (`analysis/base.py:48-58`, `analysis.py:59-66`:
    Compress, BreakLongSpans(100000), Thin(500 ln(2+n)), BinObservations(100), RecodeMonomorphic, Compress).  Measured
    with the reference end to end: the bootstrap sees 1 850 un-binned rows, the main EM loop 2 727 rows
    with 3 distinct keys, 10 (span > 1, key) groups and a longest span of 14 — reproduced exactly."""
    from smcpp_amd import data as D, vcf2smc as V
    vcf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example.vcf.gz")
    c, _ = V.vcf2smc(vcf, "1", ("pop1", ["msp_0", "msp_1", "msp_2"]))
    c.data = D.compress_repeated_obs(c.data)
    pieces = D.break_long_spans(c, 100000)
    assert len(pieces) == 1 and pieces[0].data.shape == (1850, 4)
    p = pieces[0]
    thinning = int(500 * np.log(2 + p.n[0]))
    assert thinning == 895
    b = D.bin_observations(D.thin_data(p.data, thinning), 100, [2])
    q = D.recode_monomorphic(D.Contig(b, p.pid, p.n, p.a))
    z = D.compress_repeated_obs(q.data)
    assert len(z) == 2727
    assert len(np.unique(z[:, 1:], axis=0)) == 3
    assert len(np.unique(z[z[:, 0] > 1], axis=0)) == 10
    assert int(z[:, 0].max()) == 14
    assert abs(D.watterson_theta(pieces) - 4.02337e-4) < 1e-8


# ---- golden G11: the reference's own shaping code run on real data (tests/golden/make_golden_pipeline.py) ----
def _g11():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G11_pipeline.npz"))


def test_pipeline_rows_equal_reference_on_the_example_contig():
    """Row-for-row equality with the reference's `estimation_tools.py` / `data_filter.py` on the contig made from the
    reference's example VCF (which this repository's converter must also reproduce from the VCF)."""
    from smcpp_amd import data as D, vcf2smc as V
    g = _g11()
    vcf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example.vcf.gz")
    c, _ = V.vcf2smc(vcf, "1", ("pop1", ["msp_0", "msp_1", "msp_2"]))
    assert np.array_equal(c.data, g["ex_raw"]) and list(c.n) == g["ex_n"].tolist() and list(c.a) == g["ex_a"].tolist()
    comp = D.compress_repeated_obs(c.data)
    assert np.array_equal(comp, g["ex_compress"])
    assert np.array_equal(D.decompress_polymorphic_spans(comp), g["ex_decompress"])
    for cutoff in (100000, 5000, 2000):
        pieces = D.break_long_spans(D.Contig(comp.copy(), c.pid, c.n, c.a), cutoff)
        assert len(pieces) == int(g[f"ex_break_{cutoff}_n"])
        for i, p in enumerate(pieces):
            assert np.array_equal(p.data, g[f"ex_break_{cutoff}_{i}"]), (cutoff, i)
    for cutoff in (None, 3000):
        r = D.recode_nonseg(D.Contig(comp.copy(), c.pid, c.n, c.a), cutoff)
        assert np.array_equal(r.data, g[f"ex_recode_nonseg_{cutoff}"])
    np.testing.assert_allclose(D.watterson_theta([c]), float(g["ex_watterson"]), rtol=1e-14)
    assert np.array_equal(D.validate(D.Contig(c.data.copy(), c.pid, c.n, c.a)).data, g["ex_validate"])
    # malformed rows: the reference's test is `span <= (0 | a > A | b > nb | nb > n)` (operator precedence), i.e. a violating row of
    # span > 1 passes; G11 records which of these one-violation contigs the reference's Validate refused
    for row, raised in zip(g["validate_cases"], g["validate_raised"]):
        d = np.array([[3, 0, 0, 0], row, [2, 1, 1, 4]], dtype=np.int32)
        cc = D.Contig(np.ascontiguousarray(d), ("pop1",), np.array([4]), np.array([2]))
        if raised:
            with pytest.raises(RuntimeError, match="data validation failed"):
                D.validate(cc)
        else:
            D.validate(cc)
    assert np.array_equal(D.recode_monomorphic(D.Contig(c.data.copy(), c.pid, c.n, c.a)).data, g["ex_recode_mono"])


def test_pipeline_rows_equal_reference_on_its_test_data(tmp_path):
    """The reference's own `.smc.gz` test files (test/bugs/11): reader, compression, span breaking, long-run recoding,
    Watterson's estimator and monomorphic recoding, row for row."""
    from smcpp_amd import data as D
    g = _g11()
    for fi, name in enumerate(g["bug11_files"].tolist()):
        fn = str(tmp_path / name)
        with gzip.open(fn, "wt") as f:
            f.write(str(g[f"bug11_{fi}_text"]))
        c = D.load_smc(fn)
        assert c.pid == tuple(g[f"bug11_{fi}_pid"].tolist())
        assert list(c.n) == g[f"bug11_{fi}_n"].tolist() and list(c.a) == g[f"bug11_{fi}_a"].tolist()
        d = g[f"bug11_{fi}_data"]
        assert np.array_equal(c.data, d)
        comp = D.compress_repeated_obs(d)
        assert np.array_equal(comp, g[f"bug11_{fi}_compress"])
        pieces = D.break_long_spans(D.Contig(comp.copy(), c.pid, c.n, c.a), 20000)
        assert len(pieces) == int(g[f"bug11_{fi}_break_n"])
        for i, p in enumerate(pieces):
            assert np.array_equal(p.data, g[f"bug11_{fi}_break_{i}"])
        r = D.recode_nonseg(D.Contig(comp.copy(), c.pid, c.n, c.a), 10000)
        assert np.array_equal(r.data, g[f"bug11_{fi}_recode_nonseg"])
        np.testing.assert_allclose(D.watterson_theta([D.Contig(d.copy(), c.pid, c.n, c.a)]), float(g[f"bug11_{fi}_watterson"]),
                                   rtol=1e-14)
        assert np.array_equal(D.recode_monomorphic(D.Contig(d.copy(), c.pid, c.n, c.a)).data, g[f"bug11_{fi}_recode_mono"])
    assert np.array_equal(D.compress_repeated_obs(g["kat_compress_in"]), g["kat_compress_out"])


def test_realign_puts_a_boundary_on_every_multiple_of_w():
    """`realign` (_estimation_tools.pyx:176-209; since round 6 also compared row for row with the reference's own compiled Cython:
    golden G23, test_thin_bin_realign_window_counts_vs_the_reference_cython): spans are conserved, every row keeps its observation, no
    row straddles a multiple of w counted from the last split, and a hand-worked case."""
    from smcpp_amd import data as D
    d = np.array([[5, 0, 0, 0], [1, 1, 2, 4], [7, -1, 0, 0], [3, 0, 1, 4]], dtype=np.int32)
    r = D.realign(d, 4)
    # 5 -> 4 + 1; then seen = 1, +1 = 2, then 7: 2 + 7 > 4 -> 2 (boundary), 5 -> 4 (boundary: 0 + 5 > 4), 1; then 3: 1 + 3 = 4 not > 4
    assert r.tolist() == [[4, 0, 0, 0], [1, 0, 0, 0], [1, 1, 2, 4], [2, -1, 0, 0], [4, -1, 0, 0], [1, -1, 0, 0], [3, 0, 1, 4]]
    rng = np.random.default_rng(3)
    raw = np.column_stack([rng.integers(1, 40, 500), rng.integers(-1, 2, 500), rng.integers(0, 3, 500), np.full(500, 4)]).astype(np.int32)
    for w in (1, 7, 100):
        r = D.realign(raw, w)
        assert r[:, 0].sum() == raw[:, 0].sum() and np.all(r[:, 0] > 0) and np.all(r[:, 0] <= max(w, raw[:, 0].max()))
        # expanding both to one row per base pair gives the same sequence of observations
        assert np.array_equal(np.repeat(r[:, 1:], r[:, 0], axis=0), np.repeat(raw[:, 1:], raw[:, 0], axis=0))
        if w == 1:
            assert np.all(r[:, 0] == 1)


def test_beta_kernel_density_estimate():
    """`beta_de_avg_pdf` (_estimation_tools.pyx:258-273; restated, unpinned by execution for the same reason): equals the
    average of scipy's Beta(1 + y / h, 1 + (1 - y) / h) densities over the sample, integrates to one over the sample space when
    the sample is uniform, boundary points count only where the kernel is finite there."""
    from scipy import stats
    from smcpp_amd import data as D
    rng = np.random.default_rng(0)
    X = rng.random(200)
    y = np.linspace(0.0, 1.0, 21)
    h = 0.05
    got = D.beta_de_avg_pdf(X, y, h)
    ref = np.array([stats.beta(1 + yy / h, 1 + (1 - yy) / h).pdf(X).mean() for yy in y])
    np.testing.assert_allclose(got, ref, rtol=1e-12)
    Xb = np.array([0.0, 1.0, 0.5])
    g = D.beta_de_avg_pdf(Xb, np.array([0.0, 1.0, 0.3]), 0.1)
    # y = 0: a = 1, the kernel is finite at X = 0 (value b = 11) and zero at X = 1; y = 1: the mirror image
    assert abs(g[0] - (11.0 + stats.beta(1, 11).pdf(0.5)) / 3) < 1e-12 and abs(g[1] - g[0]) < 1e-12
    assert abs(g[2] - stats.beta(4, 8).pdf(0.5) / 3) < 1e-12


def _g23_cases():
    import zlib  # noqa: F401
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "G23_estimation_tools.npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    return z, names


def _g23_expect(z, key, got):
    """Compare `got` with the golden entry `key` bit for bit (small outputs are stored whole, large ones as shape + CRC-32 of the
    int32 bytes + first / last 500 rows)."""
    import zlib
    got = np.ascontiguousarray(got, dtype=np.int32)
    if key in z.files:
        assert got.shape == z[key].shape, (key, got.shape, z[key].shape)
        assert np.array_equal(got, z[key]), key
        return
    assert tuple(z[key + "__shape"]) == got.shape, (key, got.shape, tuple(z[key + "__shape"]))
    flat = got if got.shape[0] >= got.shape[1] else np.ascontiguousarray(got.T)
    assert np.array_equal(flat[:500], z[key + "__head"]), key
    assert np.array_equal(flat[-500:], z[key + "__tail"]), key
    assert zlib.crc32(got.tobytes()) == int(z[key + "__crc"]), key


def test_thin_bin_realign_window_counts_vs_the_reference_cython():
    from smcpp_amd import data as D
    """SURVEY.md 8 f-2, pinned BY EXECUTION since round 6 (golden G23, tests/golden/make_golden_estimation_tools.py): the reference's
    own `smcpp/_estimation_tools.pyx` - its text of `thin_data` (8-84), `bin_observations` (146-173), `realign` (176-209) and
    `windowed_mutation_counts` (212-255), compiled in the build container with the one GSL-dependent function left out - run on the
    example-derived contig, on the reference's un-binned test contig test/bugs/11, on seven-column two-population rows and on a
    short-span mix; `smcpp_amd.data` must reproduce every output row for row, bit for bit."""
    z, names = _g23_cases()
    n_checked = 0
    for inp in ("ex", "chr11", "twopop", "small"):
        raw = np.ascontiguousarray(z[inp + "_in"], dtype=np.int32)
        a = [int(x) for x in z[inp + "_a"]]
        npop = (raw.shape[1] - 1) // 3
        contig = D.Contig(data=raw.copy(), pid=tuple(f"pop{i + 1}" for i in range(npop)), n=[0] * npop, a=a)
        for key in names:
            if not key.startswith(inp + "_") or key.endswith(("_in", "_a")):
                continue
            op = key[len(inp) + 1:].split("_")
            if op[0] == "thin" and len(op) == 3:
                got = D.thin_data(raw.copy(), int(op[1]), int(op[2]))
            elif op[0] == "bin":
                got = D.bin_observations(raw.copy(), int(op[1]), a)
            elif op[0] == "realign":
                got = D.realign(raw.copy(), int(op[1]))
            elif op[0] == "wmc":
                got = D.windowed_mutation_counts(contig, int(op[1]))
            elif op[0] == "thin400":
                got = D.bin_observations(D.thin_data(raw.copy(), 400, 0), 1000 if inp == "chr11" else 100, a)
            else:
                raise AssertionError(key)
            _g23_expect(z, key, got)
            n_checked += 1
    assert n_checked >= 50
