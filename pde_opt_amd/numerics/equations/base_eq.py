"""Equation interfaces (pde_opt/numerics/equations/base_eq.py:11-51) plus the hooks the HIP
engine needs: what to configure and which auxiliary fields to upload."""

from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np


class BaseEquation(ABC):
    """d(state)/dt = F(state, t)."""

    @abstractmethod
    def rhs(self, state, t):
        raise NotImplementedError

    # ---- HIP engine protocol -------------------------------------------------------------
    def _engine_problem(self) -> dict:
        """kwargs for HipEngine.configure (minus dtype / batch)."""
        raise NotImplementedError(f"{type(self).__name__} has no HIP kernel")

    def _engine_upload(self, engine, t: float = 0.0) -> None:
        """auxiliary fields; ``t`` is the local time they are frozen at for this advance call."""

    _state_trailing = ()  # trailing state axes after (nx, ny); the GPE has (2,)

    def _run_rhs(self, state, t):
        """Evaluate the RHS on the GPU for a single (nx, ny) field or a (batch, nx, ny) stack."""
        from ...engine import default_engine

        a = np.asarray(state)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        npts = len(self.domain.points)  # 2, or 3 for the 3-D equations
        nd = npts + len(self._state_trailing)
        single = a.ndim == nd
        if single:
            a = a[None]
        if a.ndim != nd + 1 or tuple(a.shape[1:1 + npts]) != tuple(self.domain.points):
            raise ValueError(
                f"state shape {np.shape(state)} does not match domain points {self.domain.points}"
            )
        eng = default_engine()
        eng.configure(dtype=a.dtype, batch=a.shape[0], **self._engine_problem())
        self._engine_upload(eng, float(t))
        eng.set_state(a)
        out = eng.rhs(float(t))
        return out[0] if single else out


class TimeSplittingEquation(BaseEquation):
    """d(state)/dt = A(state, t) + B(state, t)."""

    @abstractmethod
    def A_terms(self, state, t):
        raise NotImplementedError

    @abstractmethod
    def B_terms(self, state, t):
        raise NotImplementedError
