"""Minimal mirror of the reference's model objects as far as the inference managers use them
(`smcpp/model.py:58-96` PiecewiseModel, `smcpp/observe.py` Observable): `stepwise_values()`, `s`, `for_pop(pid)`,
`dlist`, `register(observer)` and item assignment that notifies observers with "model update" — which is what makes
`im.model = m; m[3] = x` push new parameters into the engine (`_smcpp.pyx:193-199,327-332`)."""
from __future__ import annotations

import weakref

import numpy as np


class Observable:
    def __init__(self):
        self._observers = weakref.WeakSet()

    def register(self, observer):
        self._observers.add(observer)

    def update_observers(self, message, *args, **kwargs):
        for ob in list(self._observers):
            ob.update(message, *args, **kwargs)


class PiecewiseModel(Observable):
    def __init__(self, a, s, N0=1e4, pid=None):
        super().__init__()
        assert len(a) == len(s)
        self.a = np.array(a, dtype=np.float64)
        self.s = np.array(s, dtype=np.float64)
        self.N0 = N0
        self.pid = pid

    @property
    def knots(self):
        return np.cumsum(self.s)

    def stepwise_values(self):
        return self.a

    @property
    def dlist(self):
        """Derivative directions (`model.dlist`, smcpp/model.py:83-91): one per piece when `differentiable`."""
        return list(range(len(self.a))) if self.differentiable else []

    differentiable = False

    def derivative_seeds(self):
        """[K x nder] seeds d a_k / d x_j handed to `smcpp_set_params` (the `.d()` parts of the reference's ad numbers,
        `_smcpp.pyx:70-76`): identity, i.e. gradients with respect to the piece sizes themselves."""
        return np.eye(len(self.a)) if self.differentiable else None

    def for_pop(self, pop):
        assert pop == self.pid
        return self

    def __getitem__(self, it):
        return self.a[it]

    def __setitem__(self, it, x):
        self.a[it] = x
        self.update_observers("model update")


class AdPiecewiseModel(Observable):
    """Piecewise-constant model whose piece sizes are ad VARIABLES, the way the reference's models hand parameters to
    `_smcpp.pyx` (`make_params`, 66-83): `stepwise_values()` returns ad numbers, `dlist` the variables derivatives are
    taken with respect to (`smcpp/model.py:83-91`).  Used with the Cython binding `smcpp_amd._smcpp_cy`."""

    def __init__(self, a, s, N0=1e4, pid=None, differentiable=None):
        super().__init__()
        from .ad import adnumber
        assert len(a) == len(s)
        self._vars = [adnumber(float(x), tag=("a", k)) for k, x in enumerate(a)]
        self.s = np.array(s, dtype=np.float64)
        self.N0 = N0
        self.pid = pid
        self._diff = list(range(len(a))) if differentiable is None else list(differentiable)

    def stepwise_values(self):
        return list(self._vars)

    @property
    def dlist(self):
        return [self._vars[k] for k in self._diff]

    def for_pop(self, pop):
        assert pop == self.pid
        return self

    def __getitem__(self, it):
        return self._vars[it]

    def __setitem__(self, it, x):
        from .ad import adnumber
        self._vars[it] = adnumber(float(x), tag=("a", it))
        self.update_observers("model update")


class _Pieces:
    """Plain (a, s) view handed to the managers by `TwoPopulationModel.for_pop`; `seeds` are the derivative seeds of
    its pieces in the joint direction space of the two-population model (None when nothing is differentiable)."""

    def __init__(self, a, s, seeds=None):
        self.a = np.asarray(a, dtype=np.float64)
        self.s = np.asarray(s, dtype=np.float64)
        self._seeds = seeds

    def stepwise_values(self):
        return self.a

    def derivative_seeds(self):
        return self._seeds


class TwoPopulationModel(Observable):
    """Mirror of `SMCTwoPopulationModel` (smcpp/model.py:259-333) for piecewise-constant populations: population 1's
    history `model1`, population 2's private history `model2` below (more recent than) `split`; above the split both
    follow `model1`.  `for_pop(pid)` returns what `PyTwoPopInferenceManager.update` (smcpp/_smcpp.pyx:353-368) needs:
    pid of population 1 -> model1; pid of population 2 -> model2 up to the split followed by model1; `None` -> the
    distinguished model of a manager whose two distinguished lineages sit in different populations (no coalescence
    before the split: an infinite first piece of length `split`, then model1)."""
    NPOP = 2

    def __init__(self, model1, model2, split):
        super().__init__()
        self.model1, self.model2 = model1, model2
        self._split = float(split)
        model1.register(self)
        model2.register(self)

    @property
    def pids(self):
        return [self.model1.pid, self.model2.pid]

    @property
    def N0(self):
        return self.model1.N0

    @property
    def split(self):
        return self._split

    @split.setter
    def split(self, x):
        self._split = float(x)
        self.update_observers("model update")

    def update(self, message, *args, **kwargs):
        self.update_observers("model update")

    @property
    def differentiable(self):
        return bool(self.model1.differentiable or self.model2.differentiable)

    @property
    def dlist(self):
        return list(range(len(self.model1.a) + len(self.model2.a))) if self.differentiable else []

    def _seed_rows(self, which, idx):
        """Rows of the joint seed matrix [K1 + K2 directions] for pieces `idx` of model `which` (0 or 1)."""
        if not self.differentiable:
            return None
        K1, K2 = len(self.model1.a), len(self.model2.a)
        out = np.zeros((len(idx), K1 + K2))
        m = (self.model1, self.model2)[which]
        if m.differentiable:
            for r, k in enumerate(idx):
                out[r, k + (K1 if which else 0)] = 1.0
        return out

    @staticmethod
    def _cut(m, split):
        """Piece boundaries of `m` with the split inserted: (cs with a final inf, index of the split in it)."""
        cs = np.concatenate([[0.0], np.cumsum(m.s)])
        cs[-1] = np.inf
        ip = int(np.searchsorted(cs, split))
        return np.insert(cs, ip, split), ip

    def for_pop(self, pid):
        m1, m2, split = self.model1, self.model2, self._split
        if pid is None:
            cs, ip = self._cut(m1, split)
            sp = np.diff(cs)
            sp[-1] = 1.0
            s = sp[ip - 1:].copy()
            s[0] = split
            idx = list(range(ip - 1, len(m1.a)))
            a = np.insert(np.asarray(m1.a, dtype=np.float64)[ip - 1:], 0, np.inf)
            seeds = self._seed_rows(0, idx)
            if seeds is not None:
                seeds = np.vstack([np.zeros((1, seeds.shape[1])), seeds])
            return _Pieces(a, s, seeds)
        i = self.pids.index(pid)
        if i == 0:
            return _Pieces(m1.a, m1.s, self._seed_rows(0, list(range(len(m1.a)))))
        # population 2: its own pieces below the split, population 1's above it.  Where the pieces are cut depends on the piece
        # LENGTHS and the split only - not on the sizes an optimiser moves - so that part is kept between calls
        key = (split, m1.s.tobytes(), m2.s.tobytes())
        st = getattr(self, "_pop2_struct", None)
        if st is None or st[0] != key:
            parts = []
            for which, m in ((1, m2), (0, m1)):
                cs, ip = self._cut(m, split)
                sp = np.diff(cs)
                sp[-1] = 1.0
                ap_idx = list(np.insert(np.arange(len(m.a)), ip, ip - 1))      # the piece containing the split is cut in two
                parts.append((which, sp, ap_idx, ip))
            (w2, sp2, idx2, ip2), (w1, sp1, idx1, ip1) = parts
            st = (key, np.concatenate([sp2[:ip2], sp1[ip1:]]), np.asarray(idx2[:ip2], dtype=np.intp), np.asarray(idx1[ip1:], dtype=np.intp))
            self._pop2_struct = st
        _, s, i2, i1 = st
        a = np.concatenate([np.asarray(m2.a, dtype=np.float64)[i2], np.asarray(m1.a, dtype=np.float64)[i1]])
        seeds = None
        if self.differentiable:
            seeds = np.vstack([self._seed_rows(1, list(i2)), self._seed_rows(0, list(i1))])
        return _Pieces(a, s.copy(), seeds)
