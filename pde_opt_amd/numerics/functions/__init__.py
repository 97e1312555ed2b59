"""Function representations used as closures (Legendre family; CNN/Mixer are out of scope)."""

from .lights import GaussianSpot, GaussianSpots
from .legendre import (
    ChemicalPotentialLegendrePolynomials,
    DiffusionLegendrePolynomials,
    LegendrePolynomialExpansion,
)

__all__ = [
    "GaussianSpot",
    "GaussianSpots",
    "LegendrePolynomialExpansion",
    "DiffusionLegendrePolynomials",
    "ChemicalPotentialLegendrePolynomials",
]
