"""Deterministic synthetic workloads.

The generator is language independent: SplitMix64 with seed ``0x5EED0000 + contig_index``,
``u = (x >> 11) * 2**-53``.  Every w=100 bp bin draws ``u1, u2``; bins with ``i % 20 == 0`` draw three more
(``u3, u4, u5``) whether or not they are used, so the n-th draw of a bin has a closed-form stream position and the
whole contig is generated with vectorised numpy.

The rows have the layout the reference's data pipeline hands to the inference manager
(`inference_manager.cpp:180-188`): int32 ``[L x 4]`` = ``(span, a, b, nb)``, run-length encoded
(`estimation_tools.py:51-60` ``compress_repeated_obs``).
"""
from __future__ import annotations

import zlib

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_at(seed: int, idx: np.ndarray) -> np.ndarray:
    """Outputs number ``idx`` (0-based) of a SplitMix64 stream started at ``seed``."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _uniform(seed: int, idx: np.ndarray) -> np.ndarray:
    return (splitmix64_at(seed, idx) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def rle_rows(bins: np.ndarray) -> np.ndarray:
    """Run-length encode equal consecutive ``(a, b, nb[, ...])`` bins into ``(span, a, b, nb[, ...])`` rows."""
    nbin = bins.shape[0]
    if nbin == 0:
        return np.zeros((0, bins.shape[1] + 1), dtype=np.int32)
    change = np.any(bins[1:] != bins[:-1], axis=1)
    starts = np.concatenate(([0], np.nonzero(change)[0] + 1))
    spans = np.diff(np.concatenate((starts, [nbin])))
    out = np.empty((len(starts), bins.shape[1] + 1), dtype=np.int32)
    out[:, 0] = spans
    out[:, 1:] = bins[starts]
    return np.ascontiguousarray(out)


def _pop_bins(seed: int, nbin: int, n: int, stream_off: int, together_second_pop: bool = False) -> np.ndarray:
    i = np.arange(nbin, dtype=np.int64)
    # stream position of u1 of bin i: 2 draws per bin + 3 extra for every earlier bin with i % 20 == 0
    base = 2 * i + 3 * ((i + 19) // 20) + stream_off
    u1 = _uniform(seed, base)
    u2 = _uniform(seed, base + 1)
    full = (i % 20) == 0
    fb = base[full]
    u3 = _uniform(seed, fb + 2)
    u4 = _uniform(seed, fb + 3)
    u5 = _uniform(seed, fb + 4)
    a = np.where(u2 < 0.08, 1, 0).astype(np.int32)
    if together_second_pop:
        a[:] = 0
    b = np.zeros(nbin, dtype=np.int32)
    nb = np.zeros(nbin, dtype=np.int32)
    bf = np.where(u3 < 0.7, 0, 1 + np.floor((n - 1) * u4 * u5)).astype(np.int32)
    b[full] = bf
    nb[full] = n
    miss = u1 < 0.002
    if not together_second_pop:
        a[miss] = -1
    b[miss] = 0
    nb[miss] = 0
    return np.stack([a, b, nb], axis=1)


def synth_contig(contig_index: int, length_bp: int, n: int, w: int = 100) -> np.ndarray:
    """One-population contig: int32 ``[L x 4]`` rows ``(span, a, b, nb)``."""
    nbin = length_bp // w
    seed = 0x5EED0000 + contig_index
    return rle_rows(_pop_bins(seed, nbin, n, 0))


def synth_contig_twopop(contig_index: int, length_bp: int, n1: int, n2: int, w: int = 100) -> np.ndarray:
    """Two-population contig ("together", a=(2,0)): int32 ``[L x 7]`` rows ``(span, a1,b1,nb1, a2,b2,nb2)``.

    Population 2 is drawn from an independent stream (seed + 2**32) with ``a2 = 0`` throughout."""
    nbin = length_bp // w
    seed = 0x5EED0000 + contig_index
    p1 = _pop_bins(seed, nbin, n1, 0)
    p2 = _pop_bins(seed + (1 << 32), nbin, n2, 0, together_second_pop=True)
    return rle_rows(np.concatenate([p1, p2], axis=1))


def synth_posterior_contig(rows: int, n: int, seed: int = 7) -> np.ndarray:
    """Un-binned rows as `smc++ posterior` sees them (smcpp/commands/posterior.py:48-111): long monomorphic runs with
    spans up to 1e5 separated by span-1 segregating sites, a sprinkling of missing stretches.  (numpy `default_rng`
    stream; golden G20 pins the rows by their crc.)"""
    rng = np.random.default_rng(seed)
    ob = np.zeros((rows, 4), dtype=np.int32)
    kind = rng.random(rows)
    seg = kind < 0.45                                    # segregating site: span 1, full SFS observation
    mis = (kind >= 0.45) & (kind < 0.50)                 # missing stretch
    mono = ~(seg | mis)
    ob[seg, 0] = 1
    ob[seg, 1] = rng.integers(0, 2, seg.sum())
    ob[seg, 3] = n
    ob[seg, 2] = np.where(ob[seg, 1] == 1, rng.integers(0, n + 1, seg.sum()), rng.integers(1, n + 1, seg.sum()))
    ob[mis, 0] = rng.integers(1, 5000, mis.sum()); ob[mis, 1] = -1
    ob[mono, 0] = np.minimum(100000, 1 + (rng.pareto(1.2, mono.sum()) * 200).astype(np.int64))
    ob[mono, 3] = n
    return ob


def contig_crc(rows: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(rows, dtype=np.int32).tobytes()) & 0xFFFFFFFF


# --- model parameters of SURVEY.md §8(d) -------------------------------------------------------------------------

def hidden_states(M: int) -> np.ndarray:
    """``[0, 0.01*(10/0.01)**((m-1)/(M-2)) for m=1..M-1, inf]``."""
    if M == 1:
        return np.array([0.0, np.inf])
    if M == 2:
        return np.array([0.0, 0.01, np.inf])
    m = np.arange(1, M)
    return np.concatenate(([0.0], 0.01 * (10.0 / 0.01) ** ((m - 1) / (M - 2)), [np.inf]))


def model_pieces(K: int = 16):
    """``a_k = 1 + 0.5 sin k``; ``s_0 = 0.01``, ``s_k = 0.01 (1.6**k - 1.6**(k-1))``."""
    k = np.arange(K)
    a = 1.0 + 0.5 * np.sin(k)
    s = np.empty(K)
    s[0] = 0.01
    s[1:] = 0.01 * (1.6 ** k[1:] - 1.6 ** (k[1:] - 1))
    return a, s


THETA = 2.5e-2
RHO = 6.25e-3
ALPHA = 1.0
POLARIZATION_ERROR = 0.5

# the 22 autosome-like lengths (Mbp) of config C3
C3_LENGTHS_MBP = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64,
                  47, 51]
