"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/pdeopt_hip.h declares; host logic (closure tracing, Domain, step plans, compat checks)."""
import os
import re

import numpy as np
import pytest

import pde_opt_amd as P
from pde_opt_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pdeopt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdeopt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pdeopt_hip.h but not exported"
    assert sorted(L._SIGNATURES) == declared, "ctypes binding table and header disagree"
    assert lib.pdeopt_abi_version() == 1


def test_struct_layout_matches_header():
    # sizes implied by the header: closure = 4*4 + 16*8, problem = 6*4 + 3*8 + 2*closure + 8
    import ctypes as C

    assert C.sizeof(L.Closure) == 16 + 128
    assert C.sizeof(L.Problem) == 24 + 24 + 2 * 144 + 8 + 144 + 16


def test_no_gpu_is_loud_not_silent():
    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(P.HipUnavailableError):
        P.HipEngine()
    dom = P.Domain((8, 8), ((0, 1), (0, 1)), "m")
    eq = P.CahnHilliard2DPeriodic(dom, 0.1, lambda c: c**3 - c, lambda c: 1.0)
    with pytest.raises(P.HipUnavailableError):
        eq.rhs(np.zeros((8, 8)), 0.0)
    with pytest.raises(P.HipUnavailableError):
        P.PDEModel(P.CahnHilliard2DPeriodic, dom, P.RK4).solve(
            dict(kappa=0.1, mu=lambda c: c, D=lambda c: 1.0), np.zeros((8, 8)), [0.0, 1.0])


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pde_opt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "np_oracle" not in src, f


def test_closure_tracing():
    d = P.as_closure(lambda c: c**3 - c)
    assert (d.kind, d.flags, d.coef) == (0, 0, (0.0, -1.0, 0.0, 1.0))
    d = P.as_closure(lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c))
    assert (d.flags, d.coef) == (1, (3.0, -6.0))
    assert P.as_closure(lambda c: np.ones_like(c)).coef == (1.0,)
    assert P.as_closure(0.15).coef == (0.15,)
    assert P.as_closure(lambda c: 0.15 * np.ones_like(c)).coef == (0.15,)
    d = P.as_closure(lambda c: np.exp(0.5 - c))
    assert (d.flags, d.coef) == (2, (0.5, -1.0))
    x = np.linspace(0.1, 0.9, 7)
    for fn in (lambda c: c * (1 - c), lambda c: 1 + c**2, lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c)):
        np.testing.assert_allclose(P.as_closure(fn)(x), fn(x), rtol=1e-14)
    with pytest.raises(P.UnsupportedClosureError):
        P.as_closure(lambda c: np.sin(c))
    with pytest.raises(P.UnsupportedClosureError):
        P.as_closure(lambda c: np.roll(c, 1))  # non-pointwise (CNN-like) closures are rejected


def test_legendre_closures_match_numpy_legval():
    """the reference's tests/test_functions.py:22-61, on the build's classes"""
    from numpy.polynomial.legendre import legval

    params = np.array([1.0, 0.5, 0.2, 0.1, -0.05, -0.02, 0.01])
    x = np.linspace(-1, 1, 20)
    np.testing.assert_allclose(P.LegendrePolynomialExpansion(params)(x), legval(x, params), rtol=1e-5, atol=1e-7)
    params = np.array([0.2, -0.1, 0.05, -0.02, 0.01, -0.005, 0.002])
    c = np.linspace(0, 1, 20)
    y = P.DiffusionLegendrePolynomials(params)(c)
    assert np.all(y > 0)
    np.testing.assert_allclose(y, np.exp(legval(2 * c - 1, params)), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(P.DiffusionLegendrePolynomials(params).closure_desc()(c), y, rtol=1e-12)
    params = np.array([0.3, 0.1, -0.2, -0.1, 0.45, -2.02, 0.01])
    np.testing.assert_allclose(P.ChemicalPotentialLegendrePolynomials(params)(c), legval(2 * c - 1, params), rtol=1e-5, atol=1e-7)
    chem = P.ChemicalPotentialLegendrePolynomials(np.array([0.3, 0.1, -0.2]), prior_fn=lambda v: 2.0 * v)
    np.testing.assert_allclose(chem(c), legval(2 * c - 1, [0.3, 0.1, -0.2]) + 2.0 * c, rtol=1e-5, atol=1e-7)
    # the reference takes any callable as prior_fn (legendre.py:56-74): polynomial priors fold exactly into the series,
    # the logit prior is a flag, their sum is both; anything else is refused loudly
    cc = np.linspace(0.05, 0.95, 19)
    np.testing.assert_allclose(chem.closure_desc()(cc), chem(cc), rtol=1e-13)
    for prior in (lambda v: v**3 - v, lambda v: np.log(v / (1 - v)) + 3 * (1 - 2 * v), lambda v: np.log(v / (1 - v))):
        ch = P.ChemicalPotentialLegendrePolynomials(np.array([0.3, 0.1, -0.2, 0.05]), prior_fn=prior)
        np.testing.assert_allclose(ch.closure_desc()(cc), ch(cc), rtol=1e-12, atol=1e-13)
    with pytest.raises(P.UnsupportedClosureError):
        P.ChemicalPotentialLegendrePolynomials(np.array([0.3, 0.1]), prior_fn=lambda v: np.sin(v)).closure_desc()


def test_domain_matches_reference_golden(golden):
    z = golden("domain_8x6.npz")
    d = P.Domain((8, 6), ((-1.0, 1.0), (0.0, 3.0)), "dimensionless")
    np.testing.assert_array_equal(np.array(d.dx), z["dx"])
    np.testing.assert_array_equal(np.array(d.L), z["L"])
    ax, fx = d.axes(), d.fft_axes()
    np.testing.assert_array_equal(ax[0], z["ax0"])
    np.testing.assert_array_equal(ax[1], z["ax1"])
    np.testing.assert_array_equal(fx[0], z["f0"])
    np.testing.assert_array_equal(fx[1], z["f1"])
    X, Y = d.mesh()
    np.testing.assert_array_equal(X, z["X"])
    np.testing.assert_array_equal(Y, z["Y"])


def test_equation_published_attributes(golden):
    z = golden("rhs_cases.npz")
    key = [k for k in z.files if k.endswith("/symbol")][0]
    tag = key.split("/")[3]
    nx, ny = (int(v) for v in tag.split("_")[0].split("x"))
    dom = P.Domain((nx, ny), ((-0.005 * nx, 0.005 * nx), (-0.005 * ny, 0.005 * ny)), "dimensionless")
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: c * (1 - c))
    np.testing.assert_allclose(eq.fourier_symbol, z[key], rtol=1e-14)
    for attr in ("fft", "ifft", "fourier_symbol"):
        assert hasattr(P.CahnHilliard2DPeriodic, attr)
    with pytest.raises(ValueError, match="Invalid derivative type"):
        P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: c, lambda c: c, derivs="spectral")


def test_solver_equation_compatibility_errors():
    from pde_opt_amd.utils import check_equation_solver_compatibility, prepare_solver_params

    check_equation_solver_compatibility(P.SemiImplicitFourierSpectral, P.CahnHilliard2DPeriodic)
    check_equation_solver_compatibility(P.StrangSplitting, P.GPE2DTSControl)
    check_equation_solver_compatibility(P.RK4, P.AllenCahn2DPeriodic)
    with pytest.raises(ValueError, match="missing required"):
        check_equation_solver_compatibility(P.StrangSplitting, P.CahnHilliard2DPeriodic)
    with pytest.raises(ValueError, match="missing required"):
        check_equation_solver_compatibility(P.SemiImplicitFourierSpectral, P.GPE2DTSControl)
    dom = P.Domain((8, 8), ((0, 1), (0, 1)), "m")
    eq = P.CahnHilliard2DPeriodic(dom, 0.1, lambda c: c, lambda c: 1.0)
    full = prepare_solver_params(P.SemiImplicitFourierSpectral, {"A": 0.5}, eq)
    assert set(full) == {"A", "fourier_symbol", "fft", "ifft"}
    P.SemiImplicitFourierSpectral(**full)


def test_constant_step_plan_matches_oracle():
    from oracle import np_oracle as O
    from pde_opt_amd.integrate import constant_step_plan

    for args in ((0.0, 0.05, 1e-4), (0.0, 1.0, 0.3), (0.0, 2e-5, 2e-7), (1.0, 1.0, 0.1), (0.0, 10.0, 5e-5)):
        assert constant_step_plan(*args) == O.constant_step_plan(*args)
