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

    def _engine_upload(self, engine, t: float = 0.0, t_end=None) -> None:
        """Auxiliary fields for integration over local times ``[t, t_end]`` (``t_end=None``: one
        right-hand-side evaluation at ``t``).  Fields that depend on time register a source the library
        calls at every substep / stage time (``HipEngine.set_aux_time_fn``); constant ones are uploaded once."""

    # names of constructor parameters whose VALUE may differ between the environments of one batch
    # (VectorPDEEnv): they travel with the environment (per-environment scalars / auxiliary fields)
    _per_env_controls: frozenset = frozenset()

    # the subset of those that are PLAIN NUMBERS the constructor stores untouched (nothing in __post_init__ derives
    # from them): an equation that differs from another in such a field only is that equation with the attribute
    # replaced (``_clone_with_scalar``) -- VectorPDEEnv builds ONE equation per step and carries the per-environment
    # values as an array instead of constructing (and closure-tracing) one dataclass per environment
    _scalar_controls: frozenset = frozenset()

    def _clone_with_scalar(self, name: str, value):
        """this equation with the plain-number field ``name`` set to ``value`` (``name`` in ``_scalar_controls``)"""
        import copy
        import types

        if name not in type(self)._scalar_controls:
            raise ValueError(f"{type(self).__name__}.{name} is not a plain scalar field")
        c = copy.copy(self)
        setattr(c, name, value)
        for k, v in self.__dict__.items():  # instance-bound methods (rhs = rhs_fd, as upstream) follow the clone
            if isinstance(v, types.MethodType) and v.__self__ is self:
                c.__dict__[k] = types.MethodType(v.__func__, c)
        return c

    @classmethod
    def _engine_upload_batch(cls, engine, eqs, t: float = 0.0, t_end=None) -> None:
        """``_engine_upload`` for a batch whose environment b is described by ``eqs[b]``"""
        eqs[0]._engine_upload(engine, t, t_end)

    def _time_dependent_rhs(self, t0: float = 0.0, t1=None) -> bool:
        """does F(state, t) depend on t over [t0, t1]?  (per-environment adaptive stepping needs an autonomous F)"""
        return False

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


def depends_on_time(fn, t: float, t_end) -> bool:
    """Does ``fn(t)`` (an array-valued function of local time) change over ``[t, t_end]``?  Probed at
    the two ends and two interior points; equations expose ``time_dependent=True/False`` to override
    (a callable that only changes inside a narrow window between the probes needs the override)."""
    if t_end is None or not t_end > t:
        return False
    ref = np.asarray(fn(t))
    for frac in (0.381966, 0.723607, 1.0):
        if not np.array_equal(np.asarray(fn(t + frac * (t_end - t))), ref):
            return True
    return False


class TimeSplittingEquation(BaseEquation):
    """d(state)/dt = A(state, t) + B(state, t)."""

    @abstractmethod
    def A_terms(self, state, t):
        raise NotImplementedError

    @abstractmethod
    def B_terms(self, state, t):
        raise NotImplementedError
