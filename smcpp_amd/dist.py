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


class ShardedInferenceManager:
    """The inference-manager surface (`E_step / loglik / logliks / Q / Q_with_gradient`, model / theta / rho / alpha,
    `_smcpp.pyx:122-308`) over contigs sharded across the ranks of a `torch.distributed` group, one process per GPU.

    The reference treats contigs as independent HMMs (one OpenMP task each, `src/inference_manager.cpp:89-94`) and
    only ever consumes sums over them (`InferenceManager::Q` 116-126, `loglik` 174-177 + `sum(llret)` in
    `_smcpp.pyx:303-308`).  Here every rank builds an ordinary manager over the contigs `lpt_shard` assigns to it,
    runs the E-step on its own GPU, and the ranks exchange ONE all-reduce(sum, fp64) of the packed statistics
    `[sum loglik | gamma0 | xisum | gamma_sums]`; afterwards `Q()` is evaluated on the reduced statistics and is
    identical on every rank, so an optimiser driven by it stays in lock-step without further communication.

    With backend "nccl" (= RCCL) the packed buffer is written by one kernel straight into the tensor that is reduced
    over xGMI; with "gloo" (CPU tests, or ranks sharing a device) it goes through the host.  Without an initialised
    process group (or world size 1) it degenerates to the local manager.

    `self.im` is this rank's LOCAL manager: after `E_step` its own `Q` / statistics getters see rank-local statistics until the
    reduced buffer has been handed back (`_ensure_unpacked`, done by this class's `Q` / `Q_with_gradient`) - query the reduced
    quantities through this wrapper, not through `self.im`.

    observations: the list of ALL contigs, identical on every rank; entries this rank does not own may be None if
    `lengths` gives every contig's row count (so a rank need not load the others' data).
    factory: callable(local_observations, device) -> manager; defaults to a one-population manager.
    """

    def __init__(self, n, observations, hidden_states, im_id, polarization_error, *, device=-1, group=None,
                 lengths=None, factory=None, always_reduce=False, a=None, direct_rccl=None):
        import torch.distributed as dist
        self._group = group
        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self._dist.get_world_size(group) if self._dist else 1
        self.rank = self._dist.get_rank(group) if self._dist else 0
        if lengths is None:
            lengths = [len(ob) for ob in observations]
        if self.world > len(lengths):
            raise RuntimeError(f"{self.world} ranks but only {len(lengths)} contigs: every rank needs at least one contig")
        self.owner = lpt_shard(lengths, self.world)
        self.mine = [i for i in range(len(lengths)) if self.owner[i] == self.rank]
        local = [observations[i] for i in self.mine]
        if any(ob is None for ob in local):
            raise RuntimeError("a contig assigned to this rank was not provided")
        if factory is None:
            from . import _smcpp
            if a is not None or (not np.isscalar(n) and len(n) == 2):
                # two populations (`PyTwoPopInferenceManager(n1, n2, a1, a2, ...)`, _smcpp.pyx:334-351): n = (n1, n2), a = (a1, a2)
                if a is None or np.isscalar(n) or len(n) != 2 or len(a) != 2:
                    raise RuntimeError("two populations need n = (n1, n2) and a = (a1, a2)")

                def factory(obs, dev):
                    return _smcpp.PyTwoPopInferenceManager(int(n[0]), int(n[1]), int(a[0]), int(a[1]), obs, hidden_states, im_id,
                                                           polarization_error, device=dev)
            else:
                def factory(obs, dev):
                    return _smcpp.PyOnePopInferenceManager(n, obs, hidden_states, im_id, polarization_error, device=dev)
        self.im = factory(local, device)
        self._device = self.im.device_index() if hasattr(self.im, "device_index") else 0
        # always_reduce: run the pack -> all-reduce -> unpack path even in a group of ONE rank (test hook: the device-buffer
        # branch of the RCCL backend on a single GPU)
        self._reduce = bool(self._dist) and (self.world > 1 or always_reduce)
        self._nccl = bool(self._dist) and self._dist.get_backend(group) == "nccl"
        self._buf = None
        self._ll_sum = None
        self._lls = None
        # RCCL path: the reduced statistics still sit in a device buffer - "direct" (the engine's own RCCL buffer) or "torch" (self._buf);
        # False: nothing pending (see _ensure_unpacked)
        self._unpack_pending = False
        self.last_local_stats = self.last_reduced_stats = None      # (host copies, kept only when `keep_stats` is set: tests)
        self.keep_stats = False
        import os as _os
        # RCCL path: issue the collective on the engine's stream (False / SMCPP_STREAM_ORDERED=0: host wait after the pack kernel)
        self.stream_ordered = _os.environ.get("SMCPP_STREAM_ORDERED", "1") not in ("", "0")
        self._ext = None
        if self._reduce:
            # global key dictionary: fixes the layout of the gamma_sums block and makes the engine prepare the
            # emission vectors of keys only other ranks' contigs hold (they enter Q through the reduced statistics)
            self.im.set_global_keys(union_keys(self.im.keys, group))
        # direct_rccl (opt-in; env SMCPP_RCCL_DIRECT=1): the ENGINE issues the all-reduce itself through RCCL's C API on its own
        # stream (pack kernel -> ncclAllReduce -> the reduced scalar into pinned host memory: no hop to a communication stream, no
        # copy engine), with a communicator of its own built from an id that rank 0 broadcasts over the torch group
        import os
        if direct_rccl is None:
            direct_rccl = os.environ.get("SMCPP_RCCL_DIRECT", "0") not in ("", "0")
        self._direct = bool(direct_rccl) and self._reduce and self._nccl
        if self._direct:
            import torch
            lib = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            lib = lib if os.path.exists(lib) else None              # (else the system's librccl.so.1)
            box = [self.im.rccl_unique_id(lib) if self.rank == 0 else None]
            self._dist.broadcast_object_list(box, src=self._dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            self.im.rccl_init(box[0], self.rank, self.world, lib)

    # ---- parameters: passed through to the local manager (every rank sets the same values) ----
    @property
    def model(self):
        return self.im.model

    @model.setter
    def model(self, m):
        self.im.model = m

    def set_raw(self, pi, T, keys, E):
        """Raw parameters; with more than one rank `keys` must cover the union of every rank's keys."""
        self.im.set_raw(pi, T, keys, E)

    theta = property(lambda self: self.im.theta, lambda self, v: setattr(self.im, "theta", v))
    rho = property(lambda self: self.im.rho, lambda self, v: setattr(self.im, "rho", v))
    alpha = property(lambda self: self.im.alpha, lambda self, v: setattr(self.im, "alpha", v))
    M = property(lambda self: self.im.M)

    @property
    def keys(self):
        """Union of every rank's keys (lexicographic)."""
        return union_keys(self.im.keys, self._group) if self._reduce else self.im.keys

    def _setup_exchange(self, torch, dev):
        """First RCCL exchange of this manager: the reduce buffer on the device the ENGINE lives on (the pack / unpack kernels
        dereference the pointer there) and - `stream_ordered` - torch's view of the engine's stream.  Whether the process group
        accepts a collective on that external stream is probed ONCE, here, with a dummy all-reduce every rank issues, and the
        ranks agree on the outcome through a second (ordinary) all-reduce: all of them take the same branch in every later
        E-step, none ever retries on its own."""
        if self._buf is not None:
            return
        self._buf = torch.empty(self.im.stats_len(), dtype=torch.float64, device=dev)
        ok = 0.0
        if self.stream_ordered:
            try:
                ext = torch.cuda.ExternalStream(int(self.im.stream()), device=dev)
                probe = torch.zeros(1, dtype=torch.float64, device=dev)
                with torch.cuda.stream(ext):
                    self._dist.all_reduce(probe, op=self._dist.ReduceOp.SUM, group=self._group)
                    float(probe.item())
                self._ext = ext
                ok = 1.0
            except Exception as ex:                                  # noqa: BLE001 - no external streams in this torch / group
                import warnings
                warnings.warn(f"stream-ordered exchange unavailable ({ex}); using the host-wait form")
        flag = torch.tensor([ok], dtype=torch.float64, device=dev)
        self._dist.all_reduce(flag, op=self._dist.ReduceOp.MIN, group=self._group)
        if float(flag.item()) < 1.0:
            self._ext = None

    def E_step(self, forward_backward_only=False):
        """Local E-step on this rank's contigs + the single all-reduce of the packed statistics."""
        self.im.E_step(forward_backward_only)
        self._lls = None
        self._unpack_pending = False
        if not self._reduce:
            self._ll_sum = float(self.im.loglik())
            return
        import torch
        if self._direct and not self.keep_stats:
            self._ll_sum = self.im.rccl_exchange()
            self._unpack_pending = "direct"
        elif self._nccl:
            dev = torch.device("cuda", self._device)
            self._setup_exchange(torch, dev)
            if self._ext is not None and not self.keep_stats:
                # (no fallback in here: whether the process group takes the external stream was settled COLLECTIVELY by the probe
                # of _setup_exchange - a rank that retried on its own would issue a collective its peers do not)
                with torch.cuda.stream(self._ext):
                    self.im.pack_stats_device(self._buf.data_ptr(), sync=False)
                    self._dist.all_reduce(self._buf, op=self._dist.ReduceOp.SUM, group=self._group)
                    self._ll_sum = float(self._buf[0].item())     # the one host wait of the exchange
            else:
                self.im.pack_stats_device(self._buf.data_ptr())       # returns after the kernel has finished
                if self.keep_stats:
                    self.last_local_stats = self._buf.cpu().numpy().copy()
                self._dist.all_reduce(self._buf, op=self._dist.ReduceOp.SUM, group=self._group)
                self._ll_sum = float(self._buf[0].item())             # synchronises the reduction
                if self.keep_stats:
                    self.last_reduced_stats = self._buf.cpu().numpy().copy()
            # the reduced statistics stay in the device buffer until Q asks for them (one kernel + a synchronisation per E-step
            # that a loglik-only caller - an evaluation loop, bench.py - never needs)
            self._unpack_pending = "torch"
        else:
            h = self.im.pack_stats()
            if self.keep_stats:
                self.last_local_stats = h.copy()
            t = torch.from_numpy(h)
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
            self._ll_sum = float(t[0].item())
            if self.keep_stats:
                self.last_reduced_stats = t.numpy().copy()
            self.im.unpack_stats(t.numpy())

    def loglik(self):
        """Sum over ALL contigs of every rank (`_smcpp.pyx:303-308`)."""
        if self._ll_sum is None:
            raise RuntimeError("no E-step has been run on this manager yet")
        return self._ll_sum

    def logliks(self):
        """Per-contig log-likelihoods in global contig order (one small extra collective, on demand)."""
        if self._lls is None:
            dev = None
            if self._nccl:
                import torch
                dev = torch.device("cuda", self._device)
            self._lls = allgather_logliks(self.im.logliks(), self.owner, device=dev, group=self._group)
        return self._lls

    def _ensure_unpacked(self):
        # dispatch on the path that PRODUCED the pending buffer (recorded by E_step), not on flags that may have been toggled since
        if self._unpack_pending == "direct":
            self.im.rccl_unpack()
        elif self._unpack_pending == "torch":
            self.im.unpack_stats_device(self._buf.data_ptr(), self._buf.numel())
        self._unpack_pending = False

    def Q(self, separate=False):
        self._ensure_unpacked()
        return self.im.Q(separate)

    def Q_with_gradient(self):
        self._ensure_unpacked()
        return self.im.Q_with_gradient()

    def last_timing(self):
        return self.im.last_timing()

    def chain_mode(self):
        return self.im.chain_mode()
