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
        """The in-kernel form.  The reference accepts ANY callable as ``prior_fn`` (legendre.py:56-74); the kernels
        carry one non-polynomial prior, ``log(c / (1 - c))``, and a polynomial prior of any degree is folded exactly
        into the series itself: ``p(c) = sum_k a_k c^k`` with ``c = (x + 1) / 2`` is a polynomial in ``x = 2c - 1``,
        i.e. a Legendre series (basis change by ``numpy.polynomial.legendre.poly2leg``), added coefficient by
        coefficient.  So ``prior_fn = lambda c: 2 c``, ``c**3 - c``, ``log(c/(1-c)) + 3 (1 - 2c)`` ... all run
        in-kernel; priors outside that (sin, exp, a CNN ...) raise ``UnsupportedClosureError``."""
        from numpy.polynomial import legendre as Lg
        from numpy.polynomial import polynomial as Pn

        from ..closures import JIT, POLY, jit_body_of

        params = np.array(self.expansion.params, dtype=np.float64)
        flags = 0
        if self.prior_fn is not None:
            prior = as_closure(self.prior_fn)
            if prior.kind == JIT or prior.kind != POLY or (prior.flags & ~LOGIT_PRIOR):
                # any other POINTWISE prior (tanh, exp, sqrt ...: legendre.py:56-74 takes any callable): the series by its
                # three-term recurrence + the prior's traced expression, compiled at run time (csrc/jit.hip);
                # non-pointwise callables (a CNN) fail in as_closure above
                series = jit_body_of(ClosureDesc(LEGENDRE, 0, tuple(float(v) for v in params)))
                assert series.endswith(" return r;")
                pbody = jit_body_of(prior)
                if not pbody.startswith("return "):  # a family prior spelled out as statements: wrap it
                    raise UnsupportedClosureError("this prior_fn combines forms the kernels do not take together")
                body = series[: -len(" return r;")] + " r += " + pbody[len("return "):-1] + "; return r;"
                expansion, pfn = self.expansion, prior
                return ClosureDesc(JIT, 0, (0.0,), source=body,
                                   host_fn=lambda c: expansion(2.0 * np.asarray(c) - 1.0) + pfn(np.asarray(c)))
            flags = prior.flags & LOGIT_PRIOR
            a = np.asarray(prior.coef, dtype=np.float64)
            if np.any(a != 0.0):
                in_x = np.zeros(1)
                half_x_plus_1 = np.array([0.5, 0.5])  # c as a polynomial in x
                for k in range(len(a) - 1, -1, -1):  # Horner in the polynomial ring: p = p * c + a_k
                    in_x = Pn.polyadd(Pn.polymul(in_x, half_x_plus_1), [a[k]])
                leg = Lg.poly2leg(in_x)
                n = max(len(params), len(leg))
                params = np.pad(params, (0, n - len(params))) + np.pad(leg, (0, n - len(leg)))
        return ClosureDesc(LEGENDRE, flags, tuple(float(v) for v in params))
