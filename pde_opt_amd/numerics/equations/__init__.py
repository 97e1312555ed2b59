"""PDE equation classes backed by the HIP engine."""

from .advection_diffusion import AdvectionDiffusion2D
from .base_eq import BaseEquation, TimeSplittingEquation
from .gross_pitaevskii import GPE2DTSControl
from .phase_field import AllenCahn2DPeriodic, CahnHilliard2DPeriodic, CahnHilliard3DPeriodic
from .smoothed_boundary import AllenCahn2DSmoothedBoundary, CahnHilliard2DSmoothedBoundary

__all__ = [
    "BaseEquation",
    "TimeSplittingEquation",
    "AllenCahn2DPeriodic",
    "CahnHilliard2DPeriodic",
    "CahnHilliard3DPeriodic",
    "AllenCahn2DSmoothedBoundary",
    "CahnHilliard2DSmoothedBoundary",
    "AdvectionDiffusion2D",
    "GPE2DTSControl",
]
