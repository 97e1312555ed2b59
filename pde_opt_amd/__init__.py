"""pde_opt_amd: MI355X-native hot path of pde_opt (PDEEnv / PDEModel.solve / equation.rhs).

The public names mirror ``pde_opt/__init__.py`` of the reference for the components on the hot
path; everything numeric runs in hand-written HIP kernels behind ``libpdeopt_hip.so``
(include/pdeopt_hip.h).  Importing the package needs neither a GPU nor the built library;
constructing an engine / environment / model does, and fails loudly otherwise.
"""

from ._lib import HipUnavailableError, PdeoptError
from .engine import HipEngine
from .integrate import Solution, diffeqsolve
from .numerics.closures import ClosureDesc, UnsupportedClosureError, as_closure, polynomial
from .numerics.domains import Domain
from .numerics.equations import (
    AdvectionDiffusion2D,
    AllenCahn2DPeriodic,
    AllenCahn2DSmoothedBoundary,
    BaseEquation,
    CahnHilliard2DPeriodic,
    CahnHilliard2DSmoothedBoundary,
    CahnHilliard3DPeriodic,
    GPE2DTSControl,
)
from .numerics.functions import (
    ChemicalPotentialLegendrePolynomials,
    DiffusionLegendrePolynomials,
    GaussianSpot,
    GaussianSpots,
    LegendrePolynomialExpansion,
)
from .numerics.shapes import Shape
from .numerics.solvers import (
    RK4,
    ConstantStepSize,
    Euler,
    PIDController,
    SaveAt,
    SemiImplicitFourierSpectral,
    StrangSplitting,
    Tsit5,
)
from .pde_env import PDEEnv, VectorPDEEnv
from .pde_model import PDEModel

__all__ = [
    "PDEModel", "PDEEnv", "VectorPDEEnv", "HipEngine", "diffeqsolve", "Solution",
    "BaseEquation", "AllenCahn2DPeriodic", "CahnHilliard2DPeriodic", "AdvectionDiffusion2D", "GPE2DTSControl",
    "AllenCahn2DSmoothedBoundary", "CahnHilliard2DSmoothedBoundary", "CahnHilliard3DPeriodic",
    "Domain", "Shape", "LegendrePolynomialExpansion", "DiffusionLegendrePolynomials", "ChemicalPotentialLegendrePolynomials",
    "GaussianSpot", "GaussianSpots",
    "SemiImplicitFourierSpectral", "StrangSplitting", "Euler", "RK4", "Tsit5",
    "ConstantStepSize", "PIDController", "SaveAt",
    "ClosureDesc", "as_closure", "polynomial", "UnsupportedClosureError", "HipUnavailableError", "PdeoptError",
]
