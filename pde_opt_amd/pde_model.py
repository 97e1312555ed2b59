"""PDEModel.solve on the HIP engine (the reference's pde_opt/pde_model.py:37-136).

``solve(parameters, y0, ts, solver_parameters, adjoint, dt0, max_steps, stepsize_controller)``
keeps the upstream signature and returns ``ys`` of shape ``(len(ts), *y0.shape)``; like upstream
it never throws on divergence (``throw=False``: NaNs are returned).  ``adjoint`` is accepted and
ignored: forward solves need no adjoint.

``train`` / ``optimize`` / ``residuals`` / ``mse`` (pde_model.py:138-551) differentiate through
the solver with diffrax adjoints and are out of scope of the hot path.
"""

from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

from .engine import HipEngine
from .integrate import diffeqsolve
from .numerics.solvers import ConstantStepSize, SaveAt
from .utils import check_equation_solver_compatibility, prepare_solver_params


class PDEModel:
    def __init__(self, equation_type, domain, solver_type, device: int = 0):
        self.equation_type = equation_type
        self.domain = domain
        self.solver_type = solver_type
        self.device = device
        check_equation_solver_compatibility(self.solver_type, self.equation_type)
        self._engine: Optional[HipEngine] = None

    def solve(
        self,
        parameters: Dict[str, Any],
        y0,
        ts,
        solver_parameters: Optional[Dict[str, Any]] = None,
        adjoint=None,
        dt0=0.000001,
        max_steps=1000000,
        stepsize_controller=None,
    ):
        equation = self.equation_type(domain=self.domain, **parameters)
        solver = self.solver_type(**prepare_solver_params(self.solver_type, solver_parameters or {}, equation))
        if self._engine is None:
            self._engine = HipEngine(self.device)
        ts = np.asarray(ts, dtype=np.float64)
        sol = diffeqsolve(
            equation, solver, t0=ts[0], t1=ts[-1], dt0=dt0, y0=y0, saveat=SaveAt(ts=ts),
            stepsize_controller=stepsize_controller or ConstantStepSize(), max_steps=max_steps,
            throw=False, engine=self._engine,
        )
        return sol.ys

    def _unsupported(self, name):
        raise NotImplementedError(
            f"PDEModel.{name} differentiates through the solver (diffrax adjoints + optimistix) and "
            "is outside the MI355X hot path built here; only the forward solve is provided"
        )

    def train(self, *a, **k):
        self._unsupported("train")

    def optimize(self, *a, **k):
        self._unsupported("optimize")

    def residuals(self, *a, **k):
        self._unsupported("residuals")

    def mse(self, *a, **k):
        self._unsupported("mse")
