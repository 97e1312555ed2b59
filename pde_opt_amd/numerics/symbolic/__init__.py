"""Exact right-hand sides from sympy for convergence tests (the role of the reference's
``pde_opt/numerics/symbolic``: ``cahn_hilliard_sym.py:15-48``, ``allen_cahn_sym.py:13-45``).

One base class differentiates a manufactured solution ``u_star(x, y, t)`` symbolically; a subclass only says what
the continuous right-hand side is.  ``u_exact(t)`` / ``rhs_exact(t)`` evaluate on the cell centres of ``domain``
(``Domain.mesh()``), which is what ``numerics.utils.testing.check_convergence`` compares ``equation.rhs`` with.
"""
from dataclasses import dataclass
from typing import Callable

import numpy as np

__all__ = [
    "BaseSymbolicEquation",
    "SymbolicAllenCahn2DPeriodic",
    "SymbolicCahnHilliard2DPeriodic",
    "SymbolicAdvectionDiffusion2D",
]


class BaseSymbolicEquation:
    """``u_exact(t)`` and ``rhs_exact(t)`` as arrays on ``domain`` (``base_sym_eq.py:11-22`` upstream)."""

    domain: object
    u_star: object

    def _rhs_expr(self, u, x, y, t):
        raise NotImplementedError

    def _compile(self):
        import sympy as sp
        from sympy.utilities.lambdify import lambdify

        x, y, t = sp.symbols("x y t", real=True)
        self._u_fn = lambdify((x, y, t), self.u_star, "numpy")
        self._rhs_fn = lambdify((x, y, t), self._rhs_expr(self.u_star, x, y, t), "numpy")

    def _on_mesh(self, fn, t):
        X, Y = self.domain.mesh()
        return np.broadcast_to(np.asarray(fn(X, Y, float(t)), dtype=np.float64), X.shape).copy()

    def u_exact(self, t: float):
        return self._on_mesh(self._u_fn, t)

    def rhs_exact(self, t: float):
        return self._on_mesh(self._rhs_fn, t)


def _laplacian(f, x, y):
    import sympy as sp

    return sp.diff(f, x, 2) + sp.diff(f, y, 2)


@dataclass
class SymbolicAllenCahn2DPeriodic(BaseSymbolicEquation):
    """``u_t = -R(u) (mu_h(u) - kappa lap u)`` (allen_cahn.py:81-84)"""

    domain: object
    kappa: float
    mu_sym: Callable
    R_sym: Callable
    u_star: object

    def __post_init__(self):
        self._compile()

    def _rhs_expr(self, u, x, y, t):
        return -self.R_sym(u) * (self.mu_sym(u) - self.kappa * _laplacian(u, x, y))


@dataclass
class SymbolicCahnHilliard2DPeriodic(BaseSymbolicEquation):
    """``u_t = div(D(u) grad(mu_h(u) - kappa lap u))`` (cahn_hilliard.py:89-109)"""

    domain: object
    kappa: float
    mu_sym: Callable
    D_sym: Callable
    u_star: object

    def __post_init__(self):
        self._compile()

    def _rhs_expr(self, u, x, y, t):
        import sympy as sp

        mu = self.mu_sym(u) - self.kappa * _laplacian(u, x, y)
        d = self.D_sym(u)
        return sp.diff(d * sp.diff(mu, x), x) + sp.diff(d * sp.diff(mu, y), y)


@dataclass
class SymbolicAdvectionDiffusion2D(BaseSymbolicEquation):
    """``u_t = -div(v u) + D lap u`` with ``velocity_sym(x, y, t) -> (vx, vy)`` sympy expressions (the
    conservative flux form the reference's notebooks imply)"""

    domain: object
    velocity_sym: Callable
    D: float
    u_star: object

    def __post_init__(self):
        self._compile()

    def _rhs_expr(self, u, x, y, t):
        import sympy as sp

        vx, vy = self.velocity_sym(x, y, t)
        return -(sp.diff(vx * u, x) + sp.diff(vy * u, y)) + self.D * _laplacian(u, x, y)
