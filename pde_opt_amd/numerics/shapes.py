"""Geometry for the smoothed-boundary equations: a binary mask smoothed into the level set ``psi`` that
``AllenCahn2DSmoothedBoundary`` / ``CahnHilliard2DSmoothedBoundary`` read from ``domain.geometry.smooth``
(the reference's ``pde_opt/numerics/shapes.py:21-203``).

The smoothing is itself a PDE solve -- Allen-Cahn relaxation of the mask with the curvature-driven part of the
motion removed (shapes.py:39-79) -- and runs on the GPU like every other solve here: the right-hand side is the
``PDEOPT_EQ_SHAPE_SMOOTH`` branch of the generic stencil kernel (csrc/stencil_generic.hpp), integrated by the
adaptive Tsit5 + PID driver (``integrate.py``) with the reference's tolerances.  The graph-Laplacian helpers
(mask eigenmodes, :81-203) are set-up code on the host (scipy), as upstream.
"""
import dataclasses
from typing import Optional, Tuple

import numpy as np

from .. import _lib as L
from .domains import Domain
from .equations.base_eq import BaseEquation


@dataclasses.dataclass
class _ShapeSmoothing(BaseEquation):
    """``u_t = 2 (c lap u + (1 - c) u_nn) - 18 u (1 - u)(1 - 2u) / eps^2`` (shapes.py:41-64), periodic."""

    domain: Domain
    epsilon: float
    curvature: float

    def _engine_problem(self):
        nx, ny = self.domain.points
        hx, hy = self.domain.dx
        # the ABI carries the curvature weight in `kappa` and epsilon in `gpe_k` (include/pdeopt_hip.h)
        return dict(equation=L.EQ_SHAPE_SMOOTH, nx=nx, ny=ny, hx=hx, hy=hy, kappa=float(self.curvature),
                    gpe_k=float(self.epsilon))

    def rhs(self, state, t):
        return self._run_rhs(state, t)


@dataclasses.dataclass
class Shape:
    """A shape given by a 0/1 array; ``smooth`` is its diffuse version in [0.001, 1] (shapes.py:21-37)."""

    binary: np.ndarray
    dx: Optional[Tuple[float, float]] = (1.0, 1.0)
    smooth_epsilon: float = 1.0
    smooth_curvature: float = 0.0
    smooth_dt: float = 0.1
    smooth_tf: float = 1.0
    engine: object = dataclasses.field(default=None, repr=False, compare=False)

    def __post_init__(self):
        s = np.array(self.smooth_shape(), dtype=np.float64)
        s[s < 0.001] = 0.001  # shapes.py:36-37
        s[s > 0.99] = 1.0
        self.smooth = s

    def smooth_equation(self) -> _ShapeSmoothing:
        binary = np.asarray(self.binary)
        if binary.ndim != 2:
            raise ValueError("Shape needs a 2-D binary array")
        nx, ny = binary.shape
        hx, hy = (float(self.dx[0]), float(self.dx[1]))
        dom = Domain((nx, ny), ((0.0, nx * hx), (0.0, ny * hy)), "dimensionless")
        return _ShapeSmoothing(dom, float(self.smooth_epsilon), float(self.smooth_curvature))

    def smooth_shape(self) -> np.ndarray:
        """Tsit5 + PID(rtol 1e-4, atol 1e-6) from the mask to ``smooth_tf`` (shapes.py:66-79)."""
        from ..integrate import diffeqsolve
        from .solvers import PIDController, SaveAt, Tsit5

        y0 = np.asarray(self.binary, dtype=np.float64)
        sol = diffeqsolve(self.smooth_equation(), Tsit5(), 0.0, float(self.smooth_tf), float(self.smooth_dt), y0,
                          saveat=SaveAt(t1=True), stepsize_controller=PIDController(rtol=1e-4, atol=1e-6),
                          max_steps=1_000_000, engine=self.engine)
        return sol.ys[-1]

    # ---- mask graph (host set-up) ------------------------------------------------------------------
    def laplacian_from_mask(self, periodic: bool = False):
        """Unnormalised 4-neighbour graph Laplacian of the cells where ``binary > 0`` (shapes.py:81-143).
        Returns ``(L, ids)``: CSR matrix over the nodes, and the node index of every cell (-1 outside)."""
        from scipy.sparse import coo_matrix, csr_matrix

        mask = np.asarray(self.binary) > 0
        ids = np.full(mask.shape, -1, dtype=np.int64)
        n = int(mask.sum())
        ids[mask] = np.arange(n, dtype=np.int64)
        if n == 0:
            return csr_matrix((0, 0)), ids
        heads, tails = [], []
        for axis in (0, 1):  # each undirected edge once: a node and its neighbour one step BACK along `axis`
            if periodic:
                both = mask & np.roll(mask, 1, axis=axis)
                heads.append(ids[both])
                tails.append(np.roll(ids, 1, axis=axis)[both])
            else:
                here = [slice(None)] * 2
                back = [slice(None)] * 2
                here[axis], back[axis] = slice(1, None), slice(None, -1)
                both = mask[tuple(here)] & mask[tuple(back)]
                heads.append(ids[tuple(here)][both])
                tails.append(ids[tuple(back)][both])
        u, v = np.concatenate(heads), np.concatenate(tails)
        degree = np.bincount(np.concatenate([u, v]), minlength=n).astype(np.float64)
        rows = np.concatenate([u, v, np.arange(n)])
        cols = np.concatenate([v, u, np.arange(n)])
        vals = np.concatenate([-np.ones(2 * u.size), degree])
        return coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr(), ids

    def get_shape_modes(self, N: Optional[int] = None):
        """The ``N`` lowest eigenmodes of the mask's graph Laplacian, scattered back onto the grid as
        ``shape_basis`` ``(nx, ny, N)`` with eigenvalues ``shape_basis_evals`` (shapes.py:145-203)."""
        import scipy.sparse.linalg

        lap, ids = self.laplacian_from_mask()
        n = lap.shape[0]
        if N is None:
            N = min(6, max(n - 1, 1))  # upstream passes k=None on to scipy's eigsh, whose default is 6 modes
        shift = max(float(lap.diagonal().mean()) if n else 1.0, 1.0) * 1e-8  # shift-invert just off the zero mode
        evals, evecs = scipy.sparse.linalg.eigsh(lap, k=N, which="LM", sigma=shift, tol=1e-8)
        basis = np.zeros(ids.shape + (N,))
        inside = ids >= 0
        basis[inside] = evecs[ids[inside]]
        self.shape_basis = basis
        self.shape_basis_evals = evals
