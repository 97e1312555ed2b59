"""Uniform cell-centred box grids (host side).

Same surface as the reference's ``Domain`` (pde_opt/numerics/domains.py:14-67): ``points``,
``box``, ``units``, ``geometry``; derived ``dx = (hi - lo) / points`` and ``L``; ``axes()`` are
cell mid-points, ``fft_axes()`` are ``fftfreq(points, dx)`` in cycles per unit length (the
equations apply the 2*pi), meshes use ``indexing="ij"`` so axis 0 is x.
Arrays are numpy (float64); the device kernels only ever need ``points`` and ``dx``.
"""

from __future__ import annotations

import dataclasses
from typing import Any, Optional, Tuple

import numpy as np


@dataclasses.dataclass
class Domain:
    points: Tuple[int, ...]
    box: Tuple[Tuple[float, float], ...]
    units: str
    geometry: Optional[Any] = None

    def __post_init__(self):
        if len(self.points) != len(self.box):
            raise ValueError("points and box must have one entry per dimension")
        self.L = tuple(hi - lo for lo, hi in self.box)
        self.dx = tuple(length / n for length, n in zip(self.L, self.points))

    # -- real space ------------------------------------------------------------------------
    def axes(self) -> Tuple[np.ndarray, ...]:
        out = []
        for (lo, hi), n, h in zip(self.box, self.points, self.dx):
            out.append(np.linspace(lo + h / 2, hi - h / 2, num=n))
        return tuple(out)

    def mesh(self) -> Tuple[np.ndarray, ...]:
        return tuple(np.meshgrid(*self.axes(), indexing="ij"))

    # -- Fourier space ---------------------------------------------------------------------
    def fft_axes(self) -> Tuple[np.ndarray, ...]:
        return tuple(np.fft.fftfreq(n, h) for n, h in zip(self.points, self.dx))

    def rfft_axes(self) -> Tuple[np.ndarray, ...]:
        return tuple(np.fft.rfftfreq(n, h) for n, h in zip(self.points, self.dx))

    def fft_mesh(self) -> Tuple[np.ndarray, ...]:
        return tuple(np.meshgrid(*self.fft_axes(), indexing="ij"))

    def rfft_mesh(self) -> Tuple[np.ndarray, ...]:
        return tuple(np.meshgrid(*self.rfft_axes(), indexing="ij"))

    def __str__(self):
        return (
            f"Domain with bounds {self.box} with units of {self.units} "
            f"and {self.points} collocation points."
        )
