"""Function representations used as closures (Legendre family; CNN/Mixer are out of scope)."""

from .legendre import (
    ChemicalPotentialLegendrePolynomials,
    DiffusionLegendrePolynomials,
    LegendrePolynomialExpansion,
)

__all__ = [
    "LegendrePolynomialExpansion",
    "DiffusionLegendrePolynomials",
    "ChemicalPotentialLegendrePolynomials",
]
