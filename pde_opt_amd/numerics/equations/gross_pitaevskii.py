"""Gross-Pitaevskii equation for the Strang split-step integrator, on the HIP engine.

Surface of the reference's ``GPE2DTSControl`` (pde_opt/numerics/equations/gross_pitaevskii.py:
19-81): fields ``domain, k, e, lights, trap_factor``; published ``dx, fft, ifft, A_term, xmesh,
ymesh, control, two_pi_i_k_2 ...``; state layout ``(N, N, 2)`` = (re, im).

    B(state, t) = -i/2 trap_factor ((1+e) X^2 + (1-e) Y^2) - i lights(t, X, Y) - i k |psi|^2

``lights(t, X, Y)`` is evaluated at the start time of EVERY Strang substep, as upstream's
``b_term = terms.vf(t0, y0, args)`` does (numerics/solvers.py:109 -> gross_pitaevskii.py:61,67-75):
a control that depends on time registers a source the library calls per substep
(``pdeopt_set_aux_time_fn``); one that does not is uploaded once.  ``time_dependent`` (new, default
``None`` = probe ``lights`` over the integration interval) forces either behaviour.

Quirk kept: upstream multiplies ``A_term`` by 0.0 (:62), so the committed
kinetic half-step is the identity.  Here ``A_term`` is caller data: by default it reproduces the
committed value (zeros); ``kinetic=True`` publishes the physical ``0.5j (2 pi i k)^2``.
"""

from __future__ import annotations

import dataclasses
from typing import Callable, Optional

import numpy as np

from ... import _lib as L
from ..domains import Domain
from ..functions.lights import GaussianSpots
from .base_eq import TimeSplittingEquation, depends_on_time
from .phase_field import keyed, spectral_table

# constants published by the reference module (gross_pitaevskii.py:12-15)
hbar = 1.05e-34
mass_Na23 = 3.8175406e-26
a0 = 5.29177210903e-11


@dataclasses.dataclass
class GPE2DTSControl(TimeSplittingEquation):
    domain: Domain
    k: float
    e: float
    lights: Callable
    trap_factor: float = 1.0
    kinetic: bool = False
    time_dependent: Optional[bool] = None
    fft = None
    ifft = None
    A_term = None
    dx = None

    _state_trailing = (2,)

    def __post_init__(self):
        if len(self.domain.points) != 2:
            raise ValueError("GPE2DTSControl needs a 2-D domain")
        self.dx = self.domain.dx[0]
        tab = spectral_table(self.domain)  # shared per grid: PDEEnv.step rebuilds the equation every step
        for name, arr in tab.items():
            if not name.startswith("_"):
                setattr(self, name, arr)
        if "mesh" not in tab:
            tab["mesh"] = self.domain.mesh()
            # keyed: PDEEnv.step rebuilds equation and solver every step; an A_term the engine already holds is
            # not uploaded again (the library would rebuild its spectral multiplier exp(A_term tau / 2))
            tab["A_kinetic"] = keyed(0.5j * tab["two_pi_i_k_2"], (tab["_serial"], "A_kinetic"))
            tab["A_zero"] = keyed(tab["A_kinetic"] * 0.0, (tab["_serial"], "A_zero"))
        self.fft = np.fft.fftn
        self.ifft = np.fft.ifftn
        self.xmesh, self.ymesh = tab["mesh"]
        self.control = lambda t: self.lights(t, self.xmesh, self.ymesh)
        self.A_term = tab["A_kinetic"] if self.kinetic else tab["A_zero"]

    def trap_potential(self) -> np.ndarray:
        """the time-independent part of V: 1/2 trap_factor ((1 + e) X^2 + (1 - e) Y^2)"""
        return 0.5 * self.trap_factor * ((1 + self.e) * self.xmesh**2 + (1 - self.e) * self.ymesh**2)

    def potential(self, t: float) -> np.ndarray:
        """V with b = -i (V + k |psi|^2): harmonic trap + control field."""
        trap = self.trap_potential()
        ctrl = np.asarray(self.control(t), dtype=np.float64)
        return trap + np.broadcast_to(ctrl, trap.shape)

    def _spots_in_kernel(self, t_end) -> bool:
        """lights is a GaussianSpots family member and may vary: the kernels evaluate it (no host sampling)"""
        return isinstance(self.lights, GaussianSpots) and self.time_dependent is not False and t_end is not None

    def _cell0(self):
        ax = self.domain.axes()
        return float(ax[0][0]), float(ax[1][0])

    def _engine_problem(self):
        nx, ny = self.domain.points
        hx, hy = self.domain.dx
        return dict(equation=L.EQ_GPE, nx=nx, ny=ny, hx=hx, hy=hy, gpe_k=float(self.k))

    _per_env_controls = frozenset({"k", "e", "lights", "trap_factor"})

    def _lights_vary(self, t, t_end) -> bool:
        if self.time_dependent is not None:
            return bool(self.time_dependent) and t_end is not None
        return depends_on_time(self.control, t, t_end)

    def _engine_upload(self, engine, t: float = 0.0, t_end=None):
        if self._spots_in_kernel(t_end):
            engine.set_aux(L.AUX_GPE_POTENTIAL, self.trap_potential())
            tab = self.lights.table(len(self.lights.spots))
            engine.set_gpe_spots(np.broadcast_to(tab, (engine.batch,) + tab.shape), *self._cell0())
            return
        engine.set_gpe_spots(None)
        if self._lights_vary(t, t_end):
            engine.set_aux_time_fn(L.AUX_GPE_POTENTIAL, self.potential)
        else:
            engine.set_aux(L.AUX_GPE_POTENTIAL, self.potential(t))

    @classmethod
    def _engine_upload_batch(cls, engine, eqs, t: float = 0.0, t_end=None):
        """Per-environment interaction strengths and potentials (``VectorPDEEnv``: the control of
        environment b is one of k, e, lights, trap_factor)."""
        eq0 = eqs[0]
        if any(e.kinetic != eq0.kinetic for e in eqs):
            raise ValueError("all environments of a batch must share A_term (the `kinetic` switch)")
        engine.set_env_gpe_k(0, [float(e.k) for e in eqs])
        same_trap = all(e.e == eq0.e and e.trap_factor == eq0.trap_factor for e in eqs)
        shared = same_trap and all(e.lights is eq0.lights for e in eqs)
        if shared:
            eq0._engine_upload(engine, t, t_end)
        elif all(e._spots_in_kernel(t_end) for e in eqs):
            # every environment steers its own spots: a few numbers per environment, evaluated in-kernel
            n = max(len(e.lights.spots) for e in eqs)
            engine.set_gpe_spots(np.stack([e.lights.table(n) for e in eqs]), *eq0._cell0())
            if same_trap:
                engine.set_aux(L.AUX_GPE_POTENTIAL, eq0.trap_potential())
            else:
                engine.set_aux(L.AUX_GPE_POTENTIAL, np.stack([e.trap_potential() for e in eqs]), per_env=True)
            return
        elif any(e._lights_vary(t, t_end) for e in eqs):
            engine.set_gpe_spots(None)
            engine.set_aux_time_fn(L.AUX_GPE_POTENTIAL, lambda tt: np.stack([e.potential(tt) for e in eqs]), per_env=True)
        else:
            engine.set_gpe_spots(None)
            engine.set_aux(L.AUX_GPE_POTENTIAL, np.stack([e.potential(t) for e in eqs]), per_env=True)

    def A_terms(self, state, t):
        return self.A_term * 0.0

    def B_terms(self, state, t):
        """Host evaluation (diagnostics only; the integrator forms b in-kernel)."""
        s = np.asarray(state)
        dens = s[..., 0] ** 2 + s[..., 1] ** 2
        b_im = -(self.potential(t) + self.k * dens)
        return np.stack([np.zeros_like(b_im), b_im], axis=-1)

    def rhs(self, state, t):
        return self.B_terms(state, t)
