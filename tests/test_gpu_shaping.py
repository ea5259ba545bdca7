"""SURVEY.md 8 f-2 on the device (smcpp_amd/csrc/shaping.hpp): `thin_data`, `bin_observations`, `compress_repeated_obs` and their
pipeline as integer HIP kernels (prefix scans of the spans + binary searches + row copies), through the C ABI
(`smcpp_dev_shape`).  Integer work: the bar is BIT-EXACT -
  * against golden G23 = the reference's own compiled `_estimation_tools.pyx` (tests/golden/make_golden_estimation_tools.py) for
    thin / bin on the example-derived contig, the reference's un-binned test contig, two-population rows and a short-span mix;
  * against golden G11 = the reference's own `compress_repeated_obs`;
  * against the host implementation (`smcpp_amd.data`, itself pinned by G23 / G11) on a 10^5-row un-binned contig, for every
    step and for the Thin -> Bin -> Compress pipeline, with size-independent properties at 10^6 rows (positions conserved,
    idempotence of compress, one row per bin)."""
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _expect(z, key, got):
    got = np.ascontiguousarray(got, dtype=np.int32)
    if key in z.files:
        assert got.shape == z[key].shape, (key, got.shape, z[key].shape)
        assert np.array_equal(got, z[key]), key
        return
    assert tuple(z[key + "__shape"]) == got.shape, (key, got.shape, tuple(z[key + "__shape"]))
    assert np.array_equal(got[:500], z[key + "__head"]), key
    assert np.array_equal(got[-500:], z[key + "__tail"]), key
    assert zlib.crc32(got.tobytes()) == int(z[key + "__crc"]), key


def test_thin_and_bin_on_the_device_vs_the_reference_cython():
    from smcpp_amd import data as D
    z = np.load(os.path.join(GOLDEN, "G23_estimation_tools.npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    n = 0
    for inp in ("ex", "chr11", "twopop", "small"):
        raw = np.ascontiguousarray(z[inp + "_in"], dtype=np.int32)
        a = [int(x) for x in z[inp + "_a"]]
        for key in names:
            if not key.startswith(inp + "_") or key.endswith(("_in", "_a")):
                continue
            op = key[len(inp) + 1:].split("_")
            if op[0] == "thin" and len(op) == 3:
                got = D.thin_data_device(raw, int(op[1]), int(op[2]))
            elif op[0] == "bin":
                got = D.bin_observations_device(raw, int(op[1]), a)
            elif op[0] == "thin400":
                got = D.bin_observations_device(D.thin_data_device(raw, 400, 0), 1000 if inp == "chr11" else 100, a)
            else:
                continue                                   # (realign / windowed_mutation_counts: host only)
            _expect(z, key, got)
            n += 1
    assert n >= 30


def test_compress_on_the_device_vs_the_reference():
    from smcpp_amd import data as D
    z = np.load(os.path.join(GOLDEN, "G11_pipeline.npz"))
    got = D.compress_repeated_obs_device(z["ex_raw"])
    assert np.array_equal(got, z["ex_compress"])
    # rows that repeat in long runs (binned data), runs across the block boundaries of the scan
    rng = np.random.default_rng(5)
    rows = np.repeat(rng.integers(0, 3, (40_000, 4)).astype(np.int32), rng.integers(1, 9, 40_000), axis=0)
    rows[:, 0] = rng.integers(1, 50, len(rows))
    assert np.array_equal(D.compress_repeated_obs_device(rows), D.compress_repeated_obs(rows))


def test_pipeline_on_the_device_vs_the_host_implementation():
    from smcpp_amd import data as D, synth
    raw = np.ascontiguousarray(synth.synth_posterior_contig(100_000, 8, seed=11), dtype=np.int32)
    for thinning, w, rows in ((400, 100, 100_000), (1000, 100, 30_000), (93, 10, 4_000)):
        raw = raw[:rows]
        t_h = D.thin_data(raw, thinning)
        t_d = D.thin_data_device(raw, thinning)
        assert np.array_equal(t_d, t_h)
        b_h = D.bin_observations(t_h, w, [2])
        assert np.array_equal(D.bin_observations_device(t_h, w, [2]), b_h)
        c_h = D.compress_repeated_obs(b_h)
        assert np.array_equal(D.compress_repeated_obs_device(b_h), c_h)
        assert np.array_equal(D.thin_bin_compress_device(raw, thinning, w, [2]), c_h)
    # thinning = 1 (every position kept), an offset, an offset beyond the window (the reference then thins every row)
    small = raw[:3000].copy(); small[:, 0] = np.minimum(small[:, 0], 37)
    for thinning, off in ((1, 0), (7, 3), (5, 9)):
        assert np.array_equal(D.thin_data_device(small, thinning, off), D.thin_data(small, thinning, off)), (thinning, off)


def test_pipeline_properties_at_a_million_rows():
    from smcpp_amd import data as D, synth
    raw = np.ascontiguousarray(synth.synth_posterior_contig(1_000_000, 8, seed=7), dtype=np.int32)
    P = int(raw[:, 0].astype(np.int64).sum())
    t, ms_t = D.thin_data_device(raw, 400, timing=True)
    assert int(t[:, 0].astype(np.int64).sum()) == P and np.all(t[:, 0] > 0)
    # kept positions carry span 1 and sit exactly `thinning` positions apart
    ends = np.cumsum(t[:, 0].astype(np.int64))
    full = t[:, 3] > 0
    assert np.all(t[full, 0] == 1) and np.all(ends[full] % 400 == 0)
    b, ms_b = D.bin_observations_device(t, 100, [2], timing=True)
    assert len(b) == -(-P // 100) and np.all(b[:, 0] == 1)
    c, ms_c = D.compress_repeated_obs_device(b, timing=True)
    assert int(c[:, 0].astype(np.int64).sum()) == len(b)
    assert np.all(np.any(c[1:, 1:] != c[:-1, 1:], axis=1))                 # no two neighbours equal
    assert np.array_equal(D.compress_repeated_obs_device(c), c)            # idempotent
    p, ms_p = D.thin_bin_compress_device(raw, 400, 100, [2], timing=True)
    assert np.array_equal(p, c)
    print(f"device shaping of 10^6 un-binned rows ({P} positions): thin {ms_t:.2f} ms -> {len(t)} rows, bin {ms_b:.2f} ms -> {len(b)} rows, "
          f"compress {ms_c:.2f} ms -> {len(c)} rows; pipeline {ms_p:.2f} ms")
