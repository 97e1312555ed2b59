"""Control fields of the Gross-Pitaevskii equation that the HIP kernels evaluate themselves.

``GPE2DTSControl.lights`` is any callable ``lights(t, X, Y)`` (pde_opt/numerics/equations/
gross_pitaevskii.py:43,61); the reference calls it in every Strang substep (numerics/solvers.py:109).  An
arbitrary Python callable of time can only be sampled on the host (one field per substep through
``pdeopt_set_aux_time_fn``).  ``GaussianSpots`` is the family the RL stirring controls are made of -- laser
spots with positions and amplitudes that move linearly in time during an environment step -- described by a
handful of numbers, so the split-step kernels form ``lights(t0, x, y)`` in registers at every substep
(``pdeopt_set_gpe_spots``) and a time-dependent control costs no host round trip and no extra pass.

It is an ordinary callable as well: ``spots(t, X, Y)`` evaluates the same expression with numpy, so it can be
handed to the reference's ``GPE2DTSControl`` unchanged.
"""

from __future__ import annotations

import dataclasses
from typing import Sequence, Tuple, Union

import numpy as np

MAX_SPOTS = 4  # PDEOPT_MAX_SPOTS

Linear = Union[float, Tuple[float, float]]  # value, or (value at t = 0, rate of change)


def _lin(v: Linear) -> Tuple[float, float]:
    if isinstance(v, (tuple, list)):
        a, b = v
        return float(a), float(b)
    return float(v), 0.0


@dataclasses.dataclass(frozen=True)
class GaussianSpot:
    """``(amp0 + amp_rate t) exp(-((x - x0 - x_rate t)^2 + (y - y0 - y_rate t)^2) / (2 width^2))``"""

    amp0: float
    amp_rate: float
    x0: float
    x_rate: float
    y0: float
    y_rate: float
    width: float

    def __call__(self, t, x, y):
        dx = x - (self.x0 + self.x_rate * t)
        dy = y - (self.y0 + self.y_rate * t)
        return (self.amp0 + self.amp_rate * t) * np.exp(-(dx * dx + dy * dy) * self.inv_two_w2)

    @property
    def inv_two_w2(self) -> float:
        return 1.0 / (2.0 * self.width * self.width)

    def row(self):
        return (self.amp0, self.amp_rate, self.x0, self.x_rate, self.y0, self.y_rate, self.inv_two_w2)


class GaussianSpots:
    """A sum of up to ``MAX_SPOTS`` Gaussian light spots; ``lights(t, X, Y)`` of ``GPE2DTSControl``."""

    def __init__(self, spots: Sequence[GaussianSpot]):
        spots = tuple(spots)
        if not 1 <= len(spots) <= MAX_SPOTS:
            raise ValueError(f"1..{MAX_SPOTS} spots, got {len(spots)}")
        self.spots = spots

    @classmethod
    def single(cls, amplitude: Linear, x: Linear, y: Linear, width: float) -> "GaussianSpots":
        """one spot; every argument is a number or ``(value at t = 0, rate)``"""
        (a0, a1), (x0, x1), (y0, y1) = _lin(amplitude), _lin(x), _lin(y)
        return cls([GaussianSpot(a0, a1, x0, x1, y0, y1, float(width))])

    @classmethod
    def moving(cls, amplitude: float, start, end, duration: float, width: float) -> "GaussianSpots":
        """a spot of constant amplitude travelling from ``start = (x, y)`` to ``end`` in ``duration`` -- the
        shape of ``update_control_parameter(old, new)`` in a PDEEnv whose control is the spot position"""
        (xs, ys), (xe, ye) = start, end
        return cls([GaussianSpot(float(amplitude), 0.0, float(xs), (xe - xs) / duration, float(ys), (ye - ys) / duration,
                                 float(width))])

    def __add__(self, other: "GaussianSpots") -> "GaussianSpots":
        return GaussianSpots(self.spots + other.spots)

    def __call__(self, t, x, y):
        out = 0.0
        for s in self.spots:
            out = out + s(t, x, y)
        return out

    @property
    def time_dependent(self) -> bool:
        return any(s.amp_rate or s.x_rate or s.y_rate for s in self.spots)

    def table(self, n: int) -> np.ndarray:
        """(n, 7) rows of pdeopt_light_spot, padded with zero-amplitude spots"""
        rows = [s.row() for s in self.spots] + [(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0)] * (n - len(self.spots))
        return np.asarray(rows, dtype=np.float64)
