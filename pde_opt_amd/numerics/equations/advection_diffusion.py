"""2-D advection-diffusion  du/dt = -div(v u) + D lap u  (periodic).

NOT present in the reference package at the surveyed commit -- only stale notebook call sites
remain (``AdvectionDiffusion2D(domain, velocity_fn, D, smooth=False)``,
notebooks/run_advection_diffusion.ipynb:67-72) -- so parity for this equation is UNPINNED by
reference code.  The discretisation is defined here from the reference's
own face primitives (pde_opt/numerics/utils/derivatives.py:39-61, :8-12), in conservative flux
form so the mean is conserved to rounding like the notebook's printout (…ipynb:85-86):

    Fx[i,j] = vx(x_{i+1/2}, y_j) * (u[i,j] + u[i+1,j]) / 2        (face i+1/2)
    Fy[i,j] = vy(x_i, y_{j+1/2}) * (u[i,j] + u[i,j+1]) / 2        (face j+1/2)
    rhs     = -((Fx[i,j]-Fx[i-1,j])/hx + (Fy[i,j]-Fy[i,j-1])/hy) + D lap5(u)

``velocity_fn(t, x, y) -> (vx, vy)`` is sampled on the faces on the host and frozen for the
duration of one ``advance`` call (one environment step); pass ``time_dependent=True`` to have the
solve driver re-sample it every substep chunk.
"""

from __future__ import annotations

import dataclasses
from typing import Callable

import numpy as np

from ... import _lib as L
from ..domains import Domain
from .base_eq import BaseEquation


@dataclasses.dataclass
class AdvectionDiffusion2D(BaseEquation):
    domain: Domain
    velocity_fn: Callable
    D: float
    smooth: bool = False
    time_dependent: bool = False

    def __post_init__(self):
        if len(self.domain.points) != 2:
            raise ValueError("AdvectionDiffusion2D needs a 2-D domain")
        if self.smooth:
            raise NotImplementedError("smooth=True (smoothed-boundary variant) is out of scope")
        hx, hy = self.domain.dx
        X, Y = self.domain.mesh()
        self._xf = (X + hx / 2, Y)  # x-faces (i+1/2, j)
        self._yf = (X, Y + hy / 2)  # y-faces (i, j+1/2)

    def face_velocities(self, t: float):
        vx = np.broadcast_to(np.asarray(self.velocity_fn(t, *self._xf)[0], dtype=np.float64), self.domain.points)
        vy = np.broadcast_to(np.asarray(self.velocity_fn(t, *self._yf)[1], dtype=np.float64), self.domain.points)
        return vx, vy

    def _engine_problem(self):
        nx, ny = self.domain.points
        hx, hy = self.domain.dx
        return dict(equation=L.EQ_ADVECTION_DIFFUSION, nx=nx, ny=ny, hx=hx, hy=hy, kappa=float(self.D))

    def _engine_upload(self, engine, t: float = 0.0):
        vx, vy = self.face_velocities(t)
        engine.set_aux(L.AUX_VX_FACE, vx)
        engine.set_aux(L.AUX_VY_FACE, vy)

    def rhs(self, state, t):
        return self._run_rhs(state, t)
