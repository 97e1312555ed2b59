"""Convergence harness for ``equation.rhs`` (``pde_opt/numerics/utils/testing.py:13-60`` upstream): the
numeric right-hand side of a manufactured solution against its sympy-exact value over a sequence of grids."""
from typing import Sequence

import numpy as np


def l2_rel_err(numeric, symbolic) -> float:
    """``||numeric - symbolic||_2 / ||symbolic||_2`` (testing.py:13-17)"""
    a = np.asarray(numeric, dtype=np.float64)
    b = np.asarray(symbolic, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))


def check_convergence(numeric, symbolic, numeric_args: dict, symbolic_args: dict, Ns: Sequence[int], L: float, t: float = 0.0):
    """Spacings and relative L2 errors of ``numeric(**numeric_args).rhs(u_exact, t)`` against
    ``symbolic(**symbolic_args).rhs_exact(t)`` on ``N x N`` periodic boxes ``[-L/2, L/2)^2`` (testing.py:20-60).
    The caller's dictionaries are not modified (upstream writes ``domain`` into them)."""
    from ..domains import Domain

    dxs, errors = [], []
    for n in Ns:
        domain = Domain((int(n), int(n)), ((-L / 2, L / 2), (-L / 2, L / 2)), "dimensionless")
        eq = numeric(**{**numeric_args, "domain": domain})
        exact = symbolic(**{**symbolic_args, "domain": domain})
        errors.append(l2_rel_err(eq.rhs(exact.u_exact(t), t), exact.rhs_exact(t)))
        dxs.append(float(domain.dx[0]))
    return dxs, errors


def convergence_slope(dxs, errors) -> float:
    """least-squares slope of log(error) against log(dx): the observed order of accuracy"""
    return float(np.polyfit(np.log(np.asarray(dxs, float)), np.log(np.asarray(errors, float)), 1)[0])


def plot_convergence(dx, err, orders=(0.5, 1.0, 1.5, 2.0), anchor="min"):
    """log-log plot of the errors with dotted reference slopes (testing.py:63-96); needs matplotlib"""
    import matplotlib.pyplot as plt

    order = np.argsort(np.asarray(dx, float))
    dx, err = np.asarray(dx, float)[order], np.asarray(err, float)[order]
    fig, ax = plt.subplots()
    ax.loglog(dx, err, "o-", label="measured")
    x0, y0 = (dx[0], err[0]) if anchor == "min" else (dx[-1], err[-1])
    for q in orders:
        ax.loglog(dx[[0, -1]], y0 * (dx[[0, -1]] / x0) ** q, ":", label=f"order {q:g}")
    ax.set_xlabel("dx")
    ax.set_ylabel("relative L2 error")
    ax.set_title(f"observed order {convergence_slope(dx, err):.3f}")
    ax.legend()
    ax.grid(True, which="both", linestyle="--", alpha=0.3)
    return fig
