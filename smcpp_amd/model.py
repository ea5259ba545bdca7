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
