"""First-order forward-mode AD numbers with the surface the inference managers and the optimiser of the reference use
(`smcpp/ad`, a vendored copy of the PyPI package `ad` 1.3.2): `adnumber(x, tag)` makes a variable, arithmetic makes
functions, `.x` is the value, `.d(var)` the derivative with respect to a variable and `.d()` the (mutable) dict
variable -> derivative that `_smcpp.pyx:103-114` fills from the engine's Jacobians.  Own implementation, first
derivatives only (the reference never uses the second-order terms on this path)."""
from __future__ import annotations

import math
from numbers import Number


class ADF:
    __slots__ = ("x", "_lc", "tag", "__weakref__")

    def __init__(self, value, lc=None, tag=None):
        self.x = float(value)
        self._lc = {} if lc is None else lc
        self.tag = tag

    # ---- access ----
    def d(self, x=None):
        if x is None:
            return self._lc
        return self._lc.get(x, 0.0)

    def __float__(self):
        return self.x

    def __repr__(self):
        return f"ad({self.x})" if self.tag is None else f"ad({self.x}, {self.tag})"

    __hash__ = object.__hash__

    def __eq__(self, other):                 # value comparison, identity hashing (variables are dict keys), as in `ad`
        return self.x == (other.x if isinstance(other, ADF) else other)

    def __lt__(self, other):
        return self.x < (other.x if isinstance(other, ADF) else other)

    def __le__(self, other):
        return self.x <= (other.x if isinstance(other, ADF) else other)

    def __gt__(self, other):
        return self.x > (other.x if isinstance(other, ADF) else other)

    def __ge__(self, other):
        return self.x >= (other.x if isinstance(other, ADF) else other)

    # ---- arithmetic ----
    @staticmethod
    def _comb(a, fa, b=None, fb=0.0):
        lc = {k: fa * v for k, v in a._lc.items()}
        if b is not None:
            for k, v in b._lc.items():
                lc[k] = lc.get(k, 0.0) + fb * v
        return lc

    def __add__(self, o):
        if isinstance(o, ADF):
            return ADF(self.x + o.x, self._comb(self, 1.0, o, 1.0))
        if isinstance(o, Number):
            return ADF(self.x + o, dict(self._lc))
        return NotImplemented

    __radd__ = __add__

    def __neg__(self):
        return ADF(-self.x, self._comb(self, -1.0))

    def __pos__(self):
        return self

    def __sub__(self, o):
        return self + (-o)

    def __rsub__(self, o):
        return (-self) + o

    def __mul__(self, o):
        if isinstance(o, ADF):
            return ADF(self.x * o.x, self._comb(self, o.x, o, self.x))
        if isinstance(o, Number):
            return ADF(self.x * o, self._comb(self, float(o)))
        return NotImplemented

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, ADF):
            return ADF(self.x / o.x, self._comb(self, 1.0 / o.x, o, -self.x / (o.x * o.x)))
        if isinstance(o, Number):
            return ADF(self.x / o, self._comb(self, 1.0 / o))
        return NotImplemented

    def __rtruediv__(self, o):
        return ADF(o / self.x, self._comb(self, -o / (self.x * self.x)))

    def __pow__(self, p):
        if isinstance(p, ADF):
            v = self.x ** p.x
            return ADF(v, self._comb(self, p.x * self.x ** (p.x - 1.0), p, v * math.log(self.x)))
        return ADF(self.x ** p, self._comb(self, p * self.x ** (p - 1.0)))

    def __rpow__(self, b):
        v = b ** self.x
        return ADF(v, self._comb(self, v * math.log(b)))

    def __abs__(self):
        return self if self.x >= 0 else -self


class ADV(ADF):
    """A variable: its derivative with respect to itself is one."""
    __slots__ = ()

    def __init__(self, value, tag=None):
        super().__init__(value, None, tag)
        self._lc = {self: 1.0}


def adnumber(x, tag=None):
    if isinstance(x, ADF):
        return ADF(x.x, dict(x._lc), tag)
    if hasattr(x, "__len__"):
        return [adnumber(v, tag) for v in x]
    return ADV(x, tag)


class admath:
    """`ad.admath` subset used by the reference's optimiser (exp, log, sqrt)."""

    @staticmethod
    def exp(x):
        if isinstance(x, ADF):
            v = math.exp(x.x)
            return ADF(v, ADF._comb(x, v))
        return math.exp(x)

    @staticmethod
    def log(x):
        if isinstance(x, ADF):
            return ADF(math.log(x.x), ADF._comb(x, 1.0 / x.x))
        return math.log(x)

    @staticmethod
    def sqrt(x):
        if isinstance(x, ADF):
            v = math.sqrt(x.x)
            return ADF(v, ADF._comb(x, 0.5 / v))
        return math.sqrt(x)
