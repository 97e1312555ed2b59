"""Numerical building blocks: domains, equations, closures, integrator descriptors."""

from .domains import Domain
from .equations import (
    AdvectionDiffusion2D,
    AllenCahn2DPeriodic,
    BaseEquation,
    CahnHilliard2DPeriodic,
    GPE2DTSControl,
)
from .functions import (
    ChemicalPotentialLegendrePolynomials,
    DiffusionLegendrePolynomials,
    LegendrePolynomialExpansion,
)
from .shapes import Shape
from .solvers import (
    RK4,
    ConstantStepSize,
    Euler,
    PIDController,
    SaveAt,
    SemiImplicitFourierSpectral,
    StrangSplitting,
    Tsit5,
)
