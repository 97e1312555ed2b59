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

``velocity_fn(t, x, y) -> (vx, vy)`` is sampled on the faces on the host.  A velocity that depends on
time is re-sampled at the time of EVERY right-hand-side evaluation (each Runge-Kutta stage time), as a
diffrax ``ODETerm`` would evaluate it: the library calls back per stage (``pdeopt_set_aux_time_fn``).
``time_dependent`` (default ``None`` = probe ``velocity_fn`` over the integration interval) forces either
behaviour.
"""

from __future__ import annotations

import dataclasses
from typing import Callable, Optional

import numpy as np

from ... import _lib as L
from ..domains import Domain
from .base_eq import BaseEquation, depends_on_time


@dataclasses.dataclass
class AdvectionDiffusion2D(BaseEquation):
    domain: Domain
    velocity_fn: Callable
    D: float
    smooth: bool = False
    time_dependent: Optional[bool] = None

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

    _per_env_controls = frozenset({"D", "velocity_fn"})

    def _velocity_varies(self, t, t_end) -> bool:
        if self.time_dependent is not None:
            return bool(self.time_dependent) and t_end is not None
        return depends_on_time(lambda tt: np.stack(self.face_velocities(tt)), t, t_end)

    def _time_dependent_rhs(self, t0: float = 0.0, t1=None) -> bool:
        return self._velocity_varies(t0, t1)

    @staticmethod
    def _upload_faces(engine, face_fn, varies, t, per_env):
        """face_fn(t) -> (vx, vy); one evaluation serves both fields of a stage time"""
        if not varies:
            vx, vy = face_fn(t)
            engine.set_aux(L.AUX_VX_FACE, vx, per_env=per_env)
            engine.set_aux(L.AUX_VY_FACE, vy, per_env=per_env)
            return
        cache = {}

        def at(tt, comp):
            if cache.get("t") != tt:
                cache["t"], cache["v"] = tt, face_fn(tt)
            return cache["v"][comp]

        engine.set_aux_time_fn(L.AUX_VX_FACE, lambda tt: at(tt, 0), per_env=per_env)
        engine.set_aux_time_fn(L.AUX_VY_FACE, lambda tt: at(tt, 1), per_env=per_env)

    def _engine_upload(self, engine, t: float = 0.0, t_end=None):
        self._upload_faces(engine, self.face_velocities, self._velocity_varies(t, t_end), t, False)

    @classmethod
    def _engine_upload_batch(cls, engine, eqs, t: float = 0.0, t_end=None):
        eq0 = eqs[0]
        if all(e.velocity_fn is eq0.velocity_fn for e in eqs):
            eq0._engine_upload(engine, t, t_end)
            return

        def faces(tt):
            vs = [e.face_velocities(tt) for e in eqs]
            return np.stack([v[0] for v in vs]), np.stack([v[1] for v in vs])

        cls._upload_faces(engine, faces, any(e._velocity_varies(t, t_end) for e in eqs), t, True)

    def rhs(self, state, t):
        return self._run_rhs(state, t)
