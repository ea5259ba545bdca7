"""Multi-GPU plumbing: contigs are independent HMMs, so they shard across ranks with no
data-path collective; the only exchange is ONE all-reduce(sum, fp64) per E-step of the packed statistics

    [ sum loglik | gamma0 (M) | xisum (M*M) | gamma_sums dense (Kg*M) ]

(`smcpp_pack_stats` / `smcpp_unpack_stats` in include/smcpp_engine.h produce and consume exactly this layout).
`torch.distributed` backend "nccl" is RCCL on ROCm; the CPU tests drive the same code over "gloo".
"""
from __future__ import annotations

import numpy as np


def lpt_shard(lengths, world):
    """Longest-processing-time-first assignment of contigs to ranks by row count.
    Returns ``owner[i]`` for every contig.  (22 autosomes over 8 GPUs: max load 1.04x the mean.)"""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(len(lengths), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += int(lengths[i])
    return owner


def union_keys(local_keys, group=None):
    """Lexicographically sorted union of every rank's key list (``block_key`` order, block_key.h:47-56)."""
    import torch.distributed as dist
    mine = [tuple(int(x) for x in k) for k in np.asarray(local_keys)]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.array(sorted(set(mine)), dtype=np.int32)
    allk = [None] * dist.get_world_size(group)
    dist.all_gather_object(allk, mine, group=group)
    return np.array(sorted(set(k for ks in allk for k in ks)), dtype=np.int32)


def stats_len(M, Kg):
    return 1 + M + M * M + Kg * M


def pack_host(logliks, gamma0s, xisums, gamma_sums, gkeys):
    """Python twin of ``smcpp_pack_stats`` (used by the CPU tests and to cross-check the C implementation).
    ``gamma_sums`` is the per-contig list of ``{key: vector}`` dicts the managers expose."""
    M = len(gamma0s[0])
    index = {tuple(int(x) for x in k): i for i, k in enumerate(np.asarray(gkeys))}
    buf = np.zeros(stats_len(M, len(index)))
    gs = buf[1 + M + M * M:].reshape(len(index), M)
    for c in range(len(logliks)):
        buf[0] += logliks[c]
        buf[1:1 + M] += gamma0s[c]
        buf[1 + M:1 + M + M * M] += np.asarray(xisums[c]).reshape(-1)
        for k, v in gamma_sums[c].items():
            gs[index[tuple(int(x) for x in k)]] += v
    return buf


def unpack_host(buf, M, Kg):
    gs = buf[1 + M + M * M:].reshape(Kg, M)
    return float(buf[0]), buf[1:1 + M].copy(), buf[1 + M:1 + M + M * M].reshape(M, M).copy(), gs.copy()


def allreduce_stats(buf, device=None, group=None):
    """The single collective of an E-step.  ``buf`` is a host float64 array; returns the reduced host array."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf
    t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def q_from_stats(buf, pi, T, gkeys, E_by_key):
    """``InferenceManager::Q`` values (hmm.cpp:155-193 summed over contigs, inference_manager.cpp:116-126) from the
    reduced buffer.  ``E_by_key`` maps key -> emission vector."""
    M = len(pi)
    gkeys = np.asarray(gkeys)
    _, g0, xs, gs = unpack_host(buf, M, len(gkeys))
    q = np.zeros(4)
    q[0] = float(np.sum(np.log(pi) * g0))
    for i, k in enumerate(gkeys):
        e = E_by_key[tuple(int(x) for x in k)]
        nb = int(sum(k[2::3]))
        q[2 if nb > 0 else 1] += float(np.sum(np.log(e) * gs[i]))
    q[3] = float(np.sum(np.log(T) * xs))
    return q


def allgather_logliks(local_logliks, owner, device=None, group=None):
    """``loglik()`` keeps its per-contig vector across ranks: every rank contributes the log-likelihoods
    of the contigs it owns; the result is ordered by global contig index.  ``owner[i]`` = rank of contig i (the
    ``lpt_shard`` assignment); ``local_logliks`` follow the rank's own contigs in increasing global index."""
    import torch
    import torch.distributed as dist
    owner = np.asarray(owner)
    n = len(owner)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.asarray(local_logliks, dtype=np.float64)
    rank = dist.get_rank(group)
    mine = np.nonzero(owner == rank)[0]
    assert len(mine) == len(local_logliks)
    # a sum-all-reduce of a vector that is zero outside the owned slots is an all-gather with a fixed layout
    v = np.zeros(n)
    v[mine] = np.asarray(local_logliks, dtype=np.float64)
    t = torch.from_numpy(v)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()
