"""On-disk format and pre-HMM data shaping — the producers of the row arrays the inference
manager consumes.  Restated from the reference's Python (`smcpp/estimation_tools.py:51-60,117-167,236-267`,
`smcpp/contig.py`); integer work on the host, not on the timed path."""
from __future__ import annotations

import gzip
import json
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np


@dataclass
class Contig:
    """`smcpp/contig.py:7-27`."""
    data: np.ndarray
    pid: Tuple[str, ...] = ("pop1",)
    n: List[int] = field(default_factory=list)
    a: List[int] = field(default_factory=list)
    fn: str = ""

    @property
    def npop(self):
        return len(self.pid)

    def __len__(self):
        return int(self.data[:, 0].sum())


def load_smc(fn: str) -> Contig:
    """`.smc(.gz)` reader: header `# SMC++ {json}` with `pids / dist / undist`, then space-separated int rows
    `span a b nb [a2 b2 nb2]`.  When the distinguished pair sits in the second population (a = (0, 2)) the column blocks
    are swapped so that it comes first (`estimation_tools.py:236-267`)."""
    opener = gzip.open if fn.endswith(".gz") else open
    with opener(fn, "rt") as f:
        first = next(f).strip()
        if not first.startswith("# SMC++"):
            raise RuntimeError("Data file is not in SMC++ format: %s" % fn)
        attrs = json.loads(first[7:])
        if "pids" not in attrs:
            raise RuntimeError("Data format is too old. Re-run VCF2SMC.")
        rows = [[int(x) for x in line.split()] for line in f if line.strip() and not line.startswith("#")]
    if not rows:
        raise RuntimeError("empty dataset: %s" % fn)
    A = np.array(rows, dtype=np.int32)
    a = [len(x) for x in attrs["dist"]]
    n = [len(u) for u in attrs["undist"]]
    pid = tuple(attrs["pids"])
    if len(a) == 2 and a[0] == 0 and a[1] == 2:
        n, a, pid = n[::-1], a[::-1], pid[::-1]
        A = A[:, [0, 4, 5, 6, 1, 2, 3]]
    return Contig(pid=pid, data=np.ascontiguousarray(A, dtype=np.int32), n=n, a=a, fn=fn)


def compress_repeated_obs(dataset: np.ndarray) -> np.ndarray:
    """Merge consecutive rows with equal observations, adding their spans (`estimation_tools.py:51-60`)."""
    dataset = np.asarray(dataset)
    if len(dataset) == 0:
        return dataset.copy()
    change = np.ones(len(dataset), dtype=bool)
    change[1:] = np.any(dataset[1:, 1:] != dataset[:-1, 1:], axis=1)
    starts = np.nonzero(change)[0]
    csum = np.concatenate(([0], np.cumsum(dataset[:, 0], dtype=np.int64)))
    out = dataset[starts].copy()
    out[:, 0] = csum[np.concatenate((starts[1:], [len(dataset)]))] - csum[starts]
    return out


def decompress_polymorphic_spans(dataset: np.ndarray) -> np.ndarray:
    """Rows with span > 1 that are neither missing nor non-segregating are expanded into `span` rows of span 1
    (`estimation_tools.py:63-85`)."""
    dataset = np.asarray(dataset)
    miss = np.all(dataset[:, 1::3] == -1, axis=1) & np.all(dataset[:, 3::3] == 0, axis=1)
    nonseg = np.all(dataset[:, 1::3] == 0, axis=1) & (np.all(dataset[:, 2::3] == dataset[:, 3::3], axis=1)
                                                      | np.all(dataset[:, 2::3] == 0, axis=1))
    expand = (dataset[:, 0] > 1) & ~nonseg & ~miss
    if not expand.any():
        return dataset
    reps = np.where(expand, dataset[:, 0], 1)
    out = np.repeat(dataset, reps, axis=0)
    out[np.repeat(expand, reps), 0] = 1
    return out


def validate(contig: Contig) -> Contig:
    """`Validate` (`smcpp/data_filter.py:129-165`): at sites where every sampled haplotype carries the derived allele the
    undistinguished count b is zeroed in place (the reference's assignment to the distinguished columns hits a copy and
    has no effect; `RecodeMonomorphic` does that job later); malformed rows raise."""
    d = contig.data
    a = np.asarray(contig.a)
    n = np.asarray(contig.n)
    nonseg = ((np.all(d[:, 1::3] == a[None, :], axis=1) | np.all(d[:, 1::3] == -1, axis=1))
              & np.all(d[:, 2::3] == d[:, 3::3], axis=1) & np.any(d[:, 3::3] > 0, axis=1))
    if np.any(nonseg):
        # the reference zeroes a COPY of the distinguished columns (fancy indexing) and b in place: only b changes
        d[nonseg, 2::3] = 0
    # operator precedence of the reference (data_filter.py:146-150): `|` binds tighter than `<=`, so the test is
    # `span <= (0 | any(a > A) | any(b > nb) | any(nb > n))`: a row with span > 1 passes whatever its counts are
    bad = d[:, 0] <= (0 | np.any(d[:, 1::3] > a[None, :], axis=1) | np.any(d[:, 2::3] > d[:, 3::3], axis=1)
                      | np.any(d[:, 3::3] > n[None, :], axis=1))
    if np.any(bad):
        raise RuntimeError("data validation failed")
    return contig


def drop_small_contigs(contigs, cutoff):
    """`DropSmallContigs` (`smcpp/data_filter.py:283-297`)."""
    ret = [c for c in contigs if len(c) > cutoff]
    if not ret:
        raise RuntimeError("All contigs are <.01cM (estimated). Please double check your data.")
    return ret


def drop_uninformative_contigs(contigs):
    """`DropUninformativeContigs` (`smcpp/data_filter.py:259-280`): contigs without a single variable site are dropped."""
    ret = [c for c in contigs if ((c.data[:, 1::3].sum(axis=1) > 0) | (c.data[:, 2::3].sum(axis=1) > 0)).sum() > 0]
    if not ret:
        raise RuntimeError("No contigs have mutation data. Inference is impossible.")
    return ret


def break_long_spans(contig: Contig, span_cutoff: int) -> List[Contig]:
    """Cut a contig at missing runs of at least `span_cutoff` positions; every piece starts with one missing row
    (`estimation_tools.py:117-167`)."""
    obs = contig.data
    miss = np.zeros_like(obs[0])
    miss[0] = 1
    miss[1::3] = -1
    long_spans = np.where((obs[:, 0] >= span_cutoff) & np.all(obs[:, 1::3] == -1, axis=1)
                          & np.all(obs[:, 3::3] == 0, axis=1))[0]
    out = []
    cob = 0
    for x in long_spans.tolist() + [None]:
        out.append(Contig(data=np.ascontiguousarray(np.insert(obs[cob:x], 0, miss, 0), dtype=np.int32),
                          pid=contig.pid, fn=contig.fn, n=contig.n, a=contig.a))
        if x is not None:
            cob = x + 1
    return out


def thin_data(data: np.ndarray, thinning: int, offset: int = 0) -> np.ndarray:
    """The thinning of `_estimation_tools.pyx:8-84`: walking along the contig with a counter `i` (start `offset`),
    only the LAST position of every window of `thinning` positions keeps its full observation; every other position
    keeps the distinguished counts only (`(a, 0, 0)` per population).  Rows whose distinguished counts sum to 2 are
    recoded as non-segregating (`a = 0`), at thinned positions to an all-zero row — a quirk of the reference, whose
    `b`/`nb` scratch views are never filled, mirrored here."""
    data = np.asarray(data, dtype=np.int32)
    npop = (data.shape[1] - 1) // 3
    out = []
    i = offset
    for row in data:
        span = int(row[0])
        thin = np.zeros(3 * npop, dtype=np.int32)
        thin[0::3] = row[1::3]
        sa = int(row[1::3].sum())
        if sa == 2:
            thin[0::3] = 0
        while span > 0:
            if i < thinning and i + span >= thinning:
                if thinning - i > 1:
                    out.append(np.concatenate(([thinning - i - 1], thin)))
                if sa == 2:
                    out.append(np.concatenate(([1], np.zeros(3 * npop, dtype=np.int32))))
                else:
                    out.append(np.concatenate(([1], row[1:])))
                span -= thinning - i
                i = 0
            else:
                out.append(np.concatenate(([span], thin)))
                i += span
                break
    ret = np.array(out, dtype=np.int32).reshape(-1, data.shape[1])
    assert ret[:, 0].sum() == data[:, 0].sum()
    return ret


def bin_observations(data: np.ndarray, w: int, na) -> np.ndarray:
    """Windows of `w` positions (`_estimation_tools.pyx:101-173`): each bin is represented by the row with the largest
    observed sample size `sum_pops nb + na * (a >= 0)` (first such row; when only the distinguished pair is observed a
    segregating row wins), emitted with span 1.  `na` = distinguished lineages per population."""
    data = np.array(data, dtype=np.int32, copy=True)
    K = (data.shape[1] - 1) // 3
    na = list(na)

    def process_bin(i, j):
        max_ss, mq = -2, 0
        for q in range(i, j + 1):
            if data[q, 0] == 0:
                continue
            ss, seg = 0, 0
            for aa in range(K):
                ss += int(data[q, 3 * aa + 3]) + na[aa] * int(data[q, 3 * aa + 1] >= 0)
                seg += max(0, int(data[q, 3 * aa + 1]))
            if ss > max_ss:
                mq, max_ss = q, ss
            if max_ss == 2 and seg == 1:
                mq = q
        return np.concatenate(([1], data[mq, 1:]))

    out = []
    i = j = seen = 0
    while j < data.shape[0]:
        span = int(data[j, 0])
        if seen + span > w:
            data[j, 0] = w - seen
            out.append(process_bin(i, j))
            data[j, 0] = span - (w - seen)
            seen = 0
            i = j
        else:
            j += 1
            seen += span
    out.append(process_bin(i, j - 1))
    return np.array(out, dtype=np.int32)


def watterson_theta(contigs) -> float:
    """Watterson's estimator per base pair over a list of contigs (`smcpp/data_filter.py:300-322`): segregating span
    over sum(span * (log n + 0.5 / n + 0.57721)) with n the number of observed haplotypes of each row."""
    num = 0.0
    denom = 0.0
    for c in contigs:
        d = c.data
        spans = d[:, 0]
        seg = np.any(d[:, 1::3] >= 1, axis=1) | np.any(d[:, 2::3] > 0, axis=1)
        num += spans[seg].sum()
        sample_sizes = d[:, 3::3].sum(axis=1) + (d[:, 1::3] >= 0).sum(axis=1)
        nm = sample_sizes > 0
        ss = sample_sizes[nm].astype(np.float64)
        denom += (spans[nm] * (np.log(ss) + 0.5 / ss + 0.57721)).sum()
    return float(num / denom)


def recode_monomorphic(contig: Contig) -> Contig:
    """Rows where every sampled haplotype is derived carry no information: recode them as all-ancestral
    (`smcpp/data_filter.py:325-336`).  In place, like the reference."""
    d = contig.data
    w = np.all(d[:, 1::3] == np.asarray(contig.a), axis=1) & np.all(d[:, 2::3] == d[:, 3::3], axis=1)
    d[w, 1::3] = 0
    d[w, 2::3] = 0
    return contig


def recode_nonseg(contig: Contig, cutoff) -> Contig:
    """Long runs of homozygosity (`smcpp/estimation_tools.py:88-114`): with a cutoff they become missing data; with
    `cutoff=None` the reference only warns (threshold 50 000) and leaves the data alone."""
    if cutoff is None:
        return contig
    d = contig.data
    runs = (d[:, 0] > cutoff) & np.all(d[:, 1::3] == 0, axis=1) & np.all(d[:, 2::3] == 0, axis=1)
    d[runs, 1::3] = -1
    d[runs, 3::3] = 0
    return contig


def windowed_mutation_counts(contig: Contig, w: int) -> np.ndarray:
    """Per window of `w` base pairs: [number of non-missing positions, number of heterozygous ones]
    (`smcpp/_estimation_tools.pyx:212-255`; used to choose hidden states).  Returns an int32 array [2, L // w + 1]."""
    assert w > 0
    data = contig.data
    L = int(data[:, 0].sum())
    ret = np.zeros((L // w + 1, 2), dtype=np.int32)
    npop = (data.shape[1] - 1) // 3
    j = 0
    seen = nmiss = mut = 0
    i = 0
    last = data[0].copy()
    nrows = data.shape[0]
    while i < nrows:
        span = int(last[0])
        sp = min(span, w - seen)
        extra = seen + span - w
        seen += sp
        a = 0
        for k in range(npop):
            if last[1 + 3 * k] != -1:
                a += int(last[1 + 3 * k])
            else:
                a = -1
                break
        if a >= 0:
            mut += sp * (a % 2)
            nmiss += sp
        if extra > 0:
            last[0] = extra
            ret[j] = (nmiss, mut)
            j += 1
            nmiss = mut = seen = 0
        else:
            i += 1
            if i < nrows:
                last = data[i].copy()
    ret[j] = (nmiss, mut)
    return ret.T


def realign(data: np.ndarray, w: int) -> np.ndarray:
    """Re-cut the rows of a contig so that a row boundary falls on every multiple of `w` base pairs
    (`smcpp/_estimation_tools.pyx:176-209`, "Realign contig data to have a split every w bps").  A counter `seen` of the
    positions since the last boundary runs along the rows; a row that would carry it PAST `w` (strictly: a row that ends
    exactly on the boundary is not split, and the counter keeps running into the next row - the reference's `>`, mirrored)
    is emitted up to the boundary and its remainder re-enters the loop.  The observation columns are untouched; spans sum
    to what they summed to before."""
    assert w > 0
    data = np.asarray(data, dtype=np.int32)
    n = data.shape[0]
    out = []
    last = data[0].copy()
    i = 0
    seen = 0
    while True:
        row = last.copy()
        if seen + int(last[0]) > w:
            r = w - seen
            row[0] = r
            last[0] -= r
            seen = 0
            out.append(row)
        else:
            seen += int(last[0])
            out.append(row)
            i += 1
            if i == n:
                break
            last = data[i].copy()
    ret = np.array(out, dtype=np.int32).reshape(-1, data.shape[1])
    ret = ret[ret[:, 0] > 0]
    assert ret[:, 0].sum() == data[:, 0].sum()
    return ret


def beta_de_avg_pdf(X, y, h: float) -> np.ndarray:
    """Beta-kernel density estimate (`smcpp/_estimation_tools.pyx:258-273`): for every evaluation point y_j the average over
    the sample X of the Beta(1 + y_j / h, 1 + (1 - y_j) / h) density at X_i (Chen's boundary-free kernel on [0, 1]); sample
    points exactly on 0 / 1 contribute the density's finite boundary value when the matching shape parameter is 1, nothing
    otherwise.  ln B(a, b) = lgamma(a) + lgamma(b) - lgamma(a + b) (the reference calls gsl_sf_lnbeta)."""
    import math
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    ret = np.zeros(y.shape[0])
    inner = (X > 0.0) & (X < 1.0)
    lx, l1x = np.log(X[inner]), np.log1p(-X[inner])
    n0, n1 = int(np.sum(X == 0.0)), int(np.sum(X == 1.0))
    for j in range(y.shape[0]):
        a = 1.0 + y[j] / h
        b = 1.0 + (1.0 - y[j]) / h
        ln_B = math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b)
        s = 0.0
        if a == 1.0:
            s += n0 * math.exp(-ln_B)
        if b == 1.0:
            s += n1 * math.exp(-ln_B)
        s += float(np.sum(np.exp((a - 1.0) * lx + (b - 1.0) * l1x - ln_B)))
        ret[j] = s
    return ret / len(X)


# --- the same shaping on the device (smcpp_amd/csrc/shaping.hpp) ------------------------------------------------

def _shape_on_device(mode, data, p0=0, p1=0, na=None, timing=False):
    """`smcpp_dev_shape` + `smcpp_dev_shape_fetch`: rows int32 [L][1 + 3 P] -> rows; `timing=True` also returns the device time in
    ms with the input resident in HBM.  Fails loudly without a GPU: the functions above are the host implementation."""
    import ctypes as C
    from . import _engine as E
    data = np.ascontiguousarray(data, dtype=np.int32)
    L, ncol = data.shape
    nrows = C.c_longlong(0)
    ms = C.c_double(0.0)
    na_arr = None if na is None else np.ascontiguousarray(na, dtype=np.int64)
    na_p = None if na_arr is None else na_arr.ctypes.data_as(C.POINTER(C.c_longlong))
    E.check(E.lib().smcpp_dev_shape(int(mode), L, ncol, E.iptr(data), int(p0), int(p1), na_p, C.byref(nrows), C.byref(ms)))
    out = np.empty((nrows.value, ncol), dtype=np.int32)
    E.check(E.lib().smcpp_dev_shape_fetch(E.iptr(out)))
    return (out, ms.value) if timing else out


def thin_data_device(data, thinning, offset=0, timing=False):
    """`thin_data` (`_estimation_tools.pyx:8-84`) as device kernels: bit-exact with the host function above and with the reference."""
    return _shape_on_device(0, data, thinning, offset, timing=timing)


def bin_observations_device(data, w, na, timing=False):
    """`bin_observations` (`_estimation_tools.pyx:113-173`) as device kernels."""
    return _shape_on_device(1, data, w, 0, na=na, timing=timing)


def compress_repeated_obs_device(data, timing=False):
    """`compress_repeated_obs` (`estimation_tools.py:51-60`) as device kernels."""
    return _shape_on_device(2, data, timing=timing)


def thin_bin_compress_device(data, thinning, w, na, timing=False):
    """Thin -> Bin -> Compress (`data_filter.py:166-203`) without leaving HBM between the steps."""
    return _shape_on_device(3, data, thinning, w, na=na, timing=timing)
