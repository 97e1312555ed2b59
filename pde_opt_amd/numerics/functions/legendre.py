"""Legendre-series closures (the reference's pde_opt/numerics/functions/legendre.py:12-74).

Each object is (a) callable on host arrays, like upstream, and (b) exposes ``closure_desc()`` so
the HIP kernels evaluate the very same series in-kernel by the forward three-term recurrence.
"""

from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from ..closures import EXP_WRAP, LEGENDRE, LOGIT_PRIOR, ClosureDesc, UnsupportedClosureError, as_closure


class LegendrePolynomialExpansion:
    """``sum_n params[n] P_n(x)`` for x in [-1, 1] (legendre.py:12-34)."""

    def __init__(self, params):
        self.params = np.asarray(params, dtype=np.float64)
        self.max_degree = len(self.params) - 1

    def __call__(self, inputs):
        x = np.asarray(inputs)
        total = self.params[0] * np.ones_like(x)
        if self.max_degree >= 1:
            total = total + self.params[1] * x
        older, newer = np.ones_like(x), x
        for n in range(2, self.max_degree + 1):
            nxt = ((2 * n - 1) * x * newer - (n - 1) * older) / n
            total = total + self.params[n] * nxt
            older, newer = newer, nxt
        return total


class DiffusionLegendrePolynomials:
    """``exp(Legendre(2c - 1))``: positive mobility (legendre.py:37-53)."""

    def __init__(self, params):
        self.expansion = LegendrePolynomialExpansion(params)

    def __call__(self, inputs):
        return np.exp(self.expansion(2.0 * np.asarray(inputs) - 1.0))

    def closure_desc(self) -> ClosureDesc:
        return ClosureDesc(LEGENDRE, EXP_WRAP, tuple(float(v) for v in self.expansion.params))


class ChemicalPotentialLegendrePolynomials:
    """``Legendre(2c - 1) [+ prior_fn(c)]`` (legendre.py:56-74)."""

    def __init__(self, params, prior_fn: Optional[Callable] = None):
        self.expansion = LegendrePolynomialExpansion(params)
        self.prior_fn = prior_fn

    def __call__(self, inputs):
        c = np.asarray(inputs)
        out = self.expansion(2.0 * c - 1.0)
        if self.prior_fn is not None:
            out = out + self.prior_fn(c)
        return out

    def closure_desc(self) -> ClosureDesc:
        flags = 0
        if self.prior_fn is not None:
            prior = as_closure(self.prior_fn)
            # the in-kernel family carries exactly one prior: log(c / (1 - c))
            if not (prior.flags == LOGIT_PRIOR and all(v == 0.0 for v in prior.coef)):
                raise UnsupportedClosureError(
                    "only the logit prior log(c/(1-c)) can be combined with a Legendre chemical "
                    "potential in-kernel"
                )
            flags = LOGIT_PRIOR
        return ClosureDesc(LEGENDRE, flags, tuple(float(v) for v in self.expansion.params))
