"""Cahn-Hilliard and Allen-Cahn equations with periodic boundaries, 2-D, on the HIP engine.

Same dataclass surface as the reference (pde_opt/numerics/equations/cahn_hilliard.py:31-109,
allen_cahn.py:27-84): fields ``domain, kappa, mu, D|R, derivs``; published attributes
``kx, ky, two_pi_i_kx, ..., two_pi_i_k_2, fft, ifft`` (+ ``two_pi_i_k_4, fourier_symbol`` for CH);
``rhs(state, t)``.  ``mu`` / ``D`` / ``R`` may be any callable the closure tracer understands
(numerics/closures.py), a Legendre closure object, a number, or a ``ClosureDesc``.

``rhs`` runs on the GPU: the fused stencil kernels (csrc/stencil_*.hpp) for ``derivs="fd"``, batched
rocFFT transforms + pointwise kernels (csrc/spectral.hip) for ``derivs="fourier"``.
"""

from __future__ import annotations

import dataclasses
from typing import Any

import numpy as np

from ... import _lib as L
from ..closures import as_closure
from ..domains import Domain
from .base_eq import BaseEquation


def _spectral_attributes(eq):
    """Wave-number meshes shared by CH and AC (cahn_hilliard.py:65-73, allen_cahn.py:58-65)."""
    eq.kx, eq.ky = eq.domain.fft_mesh()
    eq.two_pi_i_kx = 2j * np.pi * eq.kx
    eq.two_pi_i_ky = 2j * np.pi * eq.ky
    eq.two_pi_i_kx_2 = eq.two_pi_i_kx**2
    eq.two_pi_i_ky_2 = eq.two_pi_i_ky**2
    eq.two_pi_i_k_2 = eq.two_pi_i_kx_2 + eq.two_pi_i_ky_2
    # published for signature compatibility with the solvers; the HIP integrators use rocFFT
    eq.fft = np.fft.fftn
    eq.ifft = np.fft.ifftn


def _select_rhs(eq):
    if eq.derivs == "fd":
        eq.rhs = eq.rhs_fd
    elif eq.derivs == "fourier":
        eq.rhs = eq.rhs_fourier
    else:
        raise ValueError(f"Invalid derivative type: {eq.derivs}")


@dataclasses.dataclass
class CahnHilliard2DPeriodic(BaseEquation):
    """du/dt = div( D(u) grad( mu_h(u) - kappa lap u ) )."""

    domain: Domain
    kappa: float
    mu: Any
    D: Any
    derivs: str = "fd"
    fft = None
    ifft = None
    fourier_symbol = None

    def rhs(self, state, t):  # replaced in __post_init__, as upstream
        raise NotImplementedError("rhs method not implemented")

    def __post_init__(self):
        if len(self.domain.points) != 2:
            raise ValueError("CahnHilliard2DPeriodic needs a 2-D domain")
        _spectral_attributes(self)
        self.two_pi_i_k_4 = self.two_pi_i_k_2**2
        self.fourier_symbol = self.kappa * self.two_pi_i_k_4
        self._mu_desc = as_closure(self.mu)
        self._mob_desc = as_closure(self.D)
        _select_rhs(self)

    def _engine_problem(self):
        nx, ny = self.domain.points
        hx, hy = self.domain.dx
        return dict(equation=L.EQ_CAHN_HILLIARD, nx=nx, ny=ny, hx=hx, hy=hy, kappa=float(self.kappa),
                    mu=self._mu_desc, mob=self._mob_desc,
                    derivs=L.DERIVS_FOURIER if self.derivs == "fourier" else L.DERIVS_FD)

    def rhs_fd(self, state, t):
        return self._run_rhs(state, t)

    def rhs_fourier(self, state, t):
        return self._run_rhs(state, t)  # 7 batched rocFFT transforms (cahn_hilliard.py:82-87)


@dataclasses.dataclass
class AllenCahn2DPeriodic(BaseEquation):
    """du/dt = -R(u) ( mu_h(u) - kappa lap u )."""

    domain: Domain
    kappa: float
    mu: Any
    R: Any
    derivs: str = "fd"

    def rhs(self, state, t):
        raise NotImplementedError("rhs method not implemented")

    def __post_init__(self):
        if len(self.domain.points) != 2:
            raise ValueError("AllenCahn2DPeriodic needs a 2-D domain")
        _spectral_attributes(self)
        self._mu_desc = as_closure(self.mu)
        self._mob_desc = as_closure(self.R)
        _select_rhs(self)

    def _engine_problem(self):
        nx, ny = self.domain.points
        hx, hy = self.domain.dx
        return dict(equation=L.EQ_ALLEN_CAHN, nx=nx, ny=ny, hx=hx, hy=hy, kappa=float(self.kappa),
                    mu=self._mu_desc, mob=self._mob_desc,
                    derivs=L.DERIVS_FOURIER if self.derivs == "fourier" else L.DERIVS_FD)

    def rhs_fd(self, state, t):
        return self._run_rhs(state, t)

    def rhs_fourier(self, state, t):
        return self._run_rhs(state, t)  # 3 batched rocFFT transforms (allen_cahn.py:74-79)
