"""A minimal EM driver over the engine (reference `smcpp/optimize/optimizers.py:61-188`):
E-step on the GPU, M-step = L-BFGS-B on -Q with the forward-mode gradients the engine returns, in log-coordinates of
the piece sizes with an optional roughness penalty (the role of `model.regularizer()`, `smcpp/model.py`).

This is the caller of the hot path, kept deliberately small: hidden-state selection, the bootstrap manager and the
observer plugins of the reference's `Analysis` are not reproduced."""
from __future__ import annotations

import numpy as np
import scipy.optimize

from . import _smcpp
from .model import PiecewiseModel


def em(contigs, n, hidden_states, a0, s, theta, rho, alpha=1.0, polarization_error=0.5, iterations=5,
       penalty=0.0, bounds=(1e-2, 1e2), device=-1, callback=None):
    """Returns `(model, logliks)`: `logliks[i]` is the log-likelihood at the parameters entering EM iteration i."""
    model = PiecewiseModel(np.array(a0, dtype=float), np.array(s, dtype=float), 1e4, "pop1")
    model.differentiable = True
    im = _smcpp.PyOnePopInferenceManager(n, contigs, hidden_states, ("pop1",), polarization_error, device=device)
    im.model = model
    im.theta = theta
    im.rho = rho
    im.alpha = alpha
    K = len(model.a)
    logliks = []

    def neg_q(x):
        model.a[:] = np.exp(x)
        model.update_observers("model update")
        q, jac = im.Q_with_gradient()
        f = -q.sum()
        g = -(jac.sum(axis=0)) * np.exp(x)          # chain rule for a = exp(x)
        if penalty > 0:
            d = np.diff(x)
            f += penalty * np.sum(d * d)
            gp = np.zeros(K)
            gp[:-1] -= 2 * penalty * d
            gp[1:] += 2 * penalty * d
            g = g + gp
        return f, g

    for it in range(iterations):
        im.E_step()
        logliks.append(im.loglik())
        if callback:
            callback(it, logliks[-1], model.a.copy())
        x0 = np.log(model.a)
        res = scipy.optimize.minimize(neg_q, x0, jac=True, method="L-BFGS-B",
                                      bounds=[(np.log(bounds[0]), np.log(bounds[1]))] * K,
                                      options={"maxiter": 50})
        model.a[:] = np.exp(res.x)
        model.update_observers("model update")
    im.E_step()
    logliks.append(im.loglik())
    return model, np.array(logliks)
