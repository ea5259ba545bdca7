"""TEST INFRASTRUCTURE ONLY — plain-Python restatement of the reference's one-population emission-table
assembly.  Never imported by the product.

Follows, line by line:
  * ``NPopInferenceManager<1>::construct_bins``        inference_manager.cpp:329-386
  * ``bin_key<1>::run``                                bin_key.h:36-64
  * ``marginalize_key<1>::run``                        marginalize_key.h:21-51 (hypergeometric lift nb -> n)
  * ``convert_monomorphic / folded_key / is_monomorphic / bk_to_map_key``  inference_manager.cpp:263-327
  * ``NPopInferenceManager<1>::recompute_emission_probs``  inference_manager.cpp:389-482
  * ``tensorSlice``                                    tensorslice.h:31-42  (column ``a*(n+1)+b``)

Inputs are the conditioned SFS after ``incorporate_theta`` (``[M,3,n+1]``) and the average coalescence times, which
come either from the compiled reference (``oracle/ref.py: prep``) or from the product's own host prep.
"""
from __future__ import annotations

import math

import numpy as np

NA = 2  # distinguished lineages of a one-population manager (inference_manager.cpp:513)


def hypergeom_pdf(k: int, n1: int, n2: int, t: int) -> float:
    """GSL ``gsl_ran_hypergeometric_pdf(k, n1, n2, t)`` = C(n1,k) C(n2,t-k) / C(n1+n2,t)."""
    if t > n1 + n2:
        t = n1 + n2
    if k > n1 or k > t:
        return 0.0
    if t > n2 and k + n2 < t:
        return 0.0
    return math.exp(_lnchoose(n1, k) + _lnchoose(n2, t - k) - _lnchoose(n1 + n2, t))


def _lnchoose(n: int, m: int) -> float:
    return math.lgamma(n + 1) - math.lgamma(m + 1) - math.lgamma(n - m + 1)


def bin_key(key, cutoff=1.0):
    a, b, nb = key
    out = set()
    if a == -1:
        for aa in range(NA + 1):
            out |= bin_key((aa, b, nb), cutoff)
    else:
        out.add((a, b, nb))
        if nb > 0 and b / nb > cutoff:
            for bb in range(int(cutoff * nb), nb + 1):
                out.add((a, bb, nb))
    return out


def marginalize_key(key, n):
    a, b, nb = key
    ret = {}
    for n1 in range(b, n + b - nb + 1):
        n2 = n - n1
        k = (a, n1, n)
        ret[k] = ret.get(k, 0.0) + hypergeom_pdf(b, n1, n2, nb)
    return ret


def is_monomorphic(k):
    return k[0] == NA and k[1] == k[2]


def convert_monomorphic(k):
    return (0, 0, k[2]) if is_monomorphic(k) else k


def folded_key(k):
    return (NA - k[0], k[2] - k[1], k[2])


def construct_bins(keys, n, polarization_error):
    ret = {}
    for bk in sorted(set(keys)):
        m = {}
        for k in sorted(bin_key(bk, 1.0)):
            for mk, p in sorted(marginalize_key(k, n).items()):
                mbk = convert_monomorphic(mk)
                m[mbk] = m.get(mbk, 0.0) + (1.0 - polarization_error) * p
                fk = folded_key(mbk)
                m[fk] = m.get(fk, 0.0) + polarization_error * p
        m2 = {}
        s = 0.0
        for k, p in sorted(m.items()):
            if p <= 0 or is_monomorphic(k):
                continue
            m2[k] = p
            s += p
        if s <= 0:
            raise RuntimeError("s<=0")
        bkpm = {}
        for k, p in sorted(m2.items()):
            mk = (k[0], k[1])
            bkpm[mk] = bkpm.get(mk, 0.0) + p / s
        ret[bk] = bkpm
    return ret


def emission_probs(keys, n, csfs_theta, avg_ct, theta, alpha, polarization_error):
    """Emission vector per distinct key, ``dict key -> [M]`` in the reference's std::map (lexicographic) order."""
    M = csfs_theta.shape[0]
    emission = csfs_theta.reshape(M, 3 * (n + 1))  # row-major flattening (a, b) -> a*(n+1)+b
    e2 = np.zeros((M, 2))
    for m in range(M):
        if math.isnan(avg_ct[m]):
            e2[m, :] = 1e-20
        else:
            le = -2.0 * alpha * theta * avg_ct[m]
            e2[m, 0] = math.exp(le)
            e2[m, 1] = -math.expm1(le)
    ukeys = sorted(set(tuple(int(x) for x in k) for k in keys))
    bins = construct_bins(ukeys, n, polarization_error)
    out = {}
    for k in ukeys:
        a, b, nb = k
        reduced = nb == 0
        miss = a == -1
        if reduced and (miss or a >= 0):
            tmp = np.ones(M) if miss else e2[:, a % 2].copy()
        else:
            tmp = np.zeros(M)
            for (aa, bb), p in bins[k].items():
                tmp += p * emission[:, aa * (n + 1) + bb]
        if tmp.max() > 1.0 or tmp.min() <= 0.0:
            raise RuntimeError("probability vector not in [0, 1]")
        out[k] = tmp
    return out


# ---------------------------------------------------------------------------------------------------------------
# generic number of populations (the P = 2 instantiation of the same templates): keys are 3P ints (a, b, nb) per
# population, the emission tensor is indexed (a_1, b_1, ..., a_P, b_P) row-major with extents (na_p + 1, n_p + 1)
# (inference_manager.cpp:263-482 with P = 2, bin_key.h:66-86, marginalize_key.h:53-79, tensorslice.h:44-58)
# ---------------------------------------------------------------------------------------------------------------
def bin_key_npop(key, na, cutoff=1.0):
    P = len(na)
    if P == 1:
        a, b, nb = key
        out = set()
        if a == -1:
            for aa in range(na[0] + 1):
                out |= bin_key_npop((aa, b, nb), na, cutoff)
        else:
            out.add((a, b, nb))
            if nb > 0 and b / nb > cutoff:
                for bb in range(int(cutoff * nb), nb + 1):
                    out.add((a, bb, nb))
        return out
    left = bin_key_npop(tuple(key[:3]), na[:1], cutoff)
    right = bin_key_npop(tuple(key[3:]), na[1:], cutoff)
    return {l + r for l in left for r in right}


def marginalize_key_npop(key, n):
    P = len(n)
    left = marginalize_key(tuple(key[:3]), n[0])
    if P == 1:
        return left
    right = marginalize_key_npop(tuple(key[3:]), n[1:])
    ret = {}
    for kl, pl in left.items():
        for kr, pr in right.items():
            ret[kl + kr] = ret.get(kl + kr, 0.0) + pl * pr
    return ret


def construct_bins_npop(keys, n, na, polarization_error):
    P = len(n)

    def is_mono(k):
        return all(k[3 * p] == na[p] and k[3 * p + 1] == k[3 * p + 2] for p in range(P))

    def conv_mono(k):
        if not is_mono(k):
            return k
        return tuple(x for p in range(P) for x in (0, 0, k[3 * p + 2]))

    def folded(k):
        return tuple(x for p in range(P) for x in (na[p] - k[3 * p], k[3 * p + 2] - k[3 * p + 1], k[3 * p + 2]))

    ret = {}
    for bk in sorted(set(keys)):
        m = {}
        for k in sorted(bin_key_npop(bk, na, 1.0)):
            for mk, p in sorted(marginalize_key_npop(k, n).items()):
                mbk = conv_mono(mk)
                m[mbk] = m.get(mbk, 0.0) + (1.0 - polarization_error) * p
                fk = folded(mbk)
                m[fk] = m.get(fk, 0.0) + polarization_error * p
        m2, s = {}, 0.0
        for k, p in sorted(m.items()):
            if p <= 0 or is_mono(k):
                continue
            m2[k] = p
            s += p
        if s <= 0:
            raise RuntimeError("s<=0")
        bkpm = {}
        for k, p in sorted(m2.items()):
            mk = tuple(x for q in range(P) for x in (k[3 * q], k[3 * q + 1]))
            bkpm[mk] = bkpm.get(mk, 0.0) + p / s
        ret[bk] = bkpm
    return ret


def incorporate_theta(csfs, theta):
    """`incorporate_theta` (conditioned_sfs.cpp:100-148) on arrays [M, ...]: scale to a probability table, put the
    remainder on the all-ancestral entry, floor at 1e-10."""
    out = []
    for c in csfs:
        c = np.array(c, dtype=np.float64)
        tauh = c.sum()
        r = c * (-math.expm1(-theta * tauh) / tauh)
        r.flat[0] = 1.0 - (r.sum())
        r = np.where(r < 1e-10, 1e-10, r)
        if r.min() < 0 or r.max() > 1:
            raise RuntimeError("csfs is not a probability distribution")
        out.append(r)
    return np.array(out)


def emission_probs_npop(keys, n, na, tensor_theta, avg_ct, theta, alpha, polarization_error):
    """Generic-P `recompute_emission_probs`: `tensor_theta` [M, na_1+1, n_1+1, ..., na_P+1, n_P+1] after
    incorporate_theta; returns dict key -> [M]."""
    P = len(n)
    M = tensor_theta.shape[0]
    e2 = np.zeros((M, 2))
    for m in range(M):
        if math.isnan(avg_ct[m]):
            e2[m, :] = 1e-20
        else:
            le = -2.0 * alpha * theta * avg_ct[m]
            e2[m, 0] = math.exp(le)
            e2[m, 1] = -math.expm1(le)
    ukeys = sorted(set(tuple(int(x) for x in k) for k in keys))
    bins = construct_bins_npop(ukeys, n, na, polarization_error)
    out = {}
    for k in ukeys:
        a = [k[3 * p] for p in range(P)]
        reduced = all(k[3 * p + 2] == 0 for p in range(P))
        miss = all(k[3 * p] == -1 for p in range(P) if na[p] > 0)
        if reduced and (miss or min(a) >= 0):
            tmp = np.ones(M) if miss else e2[:, sum(a) % 2].copy()
        else:
            tmp = np.zeros(M)
            for idx, p in bins[k].items():
                tmp += p * tensor_theta[(slice(None),) + tuple(idx)]
        if tmp.max() > 1.0 or tmp.min() <= 0.0:
            raise RuntimeError("probability vector not in [0, 1]")
        out[k] = tmp
    return out
