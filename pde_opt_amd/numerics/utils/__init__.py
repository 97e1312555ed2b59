"""Helpers around the equations (``pde_opt/numerics/utils`` upstream; the finite-difference primitives of
``derivatives.py`` live fused inside the HIP kernels, csrc/stencil_*.hpp)."""
from .testing import check_convergence, convergence_slope, l2_rel_err  # noqa: F401
