"""CahnHilliard3DPeriodic (pde_opt/numerics/equations/cahn_hilliard.py:113-200) on
the MI355X: RHS against the reference's goldens, explicit and IMEX trajectories against the oracle."""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from util import MOB, MU, TOL, rel_l2

pytestmark = pytest.mark.gpu


def _dom(nx, ny, nz):
    return P.Domain((nx, ny, nz), ((-0.005 * nx, 0.005 * nx), (-0.005 * ny, 0.005 * ny), (0.0, 0.012 * nz)), "dimensionless")


def test_rhs_against_reference_goldens(golden):
    z = golden("ch3d_cases.npz")
    tags = sorted(k[: -len("/rhs")] for k in z.files if k.endswith("/rhs"))
    assert len(tags) == 6
    for tag in tags:
        nx, ny, nz = (int(v) for v in tag.split("_")[0].split("x"))
        u, want = z[tag + "/u"], z[tag + "/rhs"]
        eq = P.CahnHilliard3DPeriodic(_dom(nx, ny, nz), 0.002, MU["regsol"], MOB["c1mc"], derivs="fd")
        got = eq.rhs(u, 0.0)
        assert got.dtype == want.dtype and got.shape == want.shape
        assert rel_l2(got, want) < TOL[want.dtype], (tag, rel_l2(got, want))
        if tag + "/symbol" in z.files:
            np.testing.assert_allclose(eq.fourier_symbol, z[tag + "/symbol"], rtol=1e-13)
    assert "CH-3D" in P.engine.default_engine().last_kernel
    # a batch equals its members
    eq = P.CahnHilliard3DPeriodic(_dom(16, 12, 20), 0.002, MU["regsol"], MOB["c1mc"])
    u = z["16x12x20_float64/u"]
    ub = np.stack([u, 0.5 * u + 0.2])
    np.testing.assert_array_equal(eq.rhs(ub, 0.0)[0], eq.rhs(u, 0.0))


@pytest.mark.parametrize("solver", ["euler", "rk4", "tsit5"])
def test_explicit_trajectory_vs_oracle(solver):
    rng = np.random.default_rng(5)
    nx, ny, nz = 24, 16, 40
    dom = _dom(nx, ny, nz)
    hx, hy, hz = dom.dx
    eq = P.CahnHilliard3DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((2, nx, ny, nz)), 0.05, 0.95)
    f = lambda t, u: O.ch3d_rhs_fd(u, hx, hy, hz, 0.002, MU["regsol"], MOB["c1mc"])
    dt, n = 1e-7, 5
    s = {"euler": P.Euler(), "rk4": P.RK4(), "tsit5": P.Tsit5()}[solver]
    sol = P.diffeqsolve(eq, s, 0.0, n * dt, dt, y0)
    for b in range(2):
        ref = y0[b]
        for i in range(n):
            ref = O.euler_step(f, 0.0, ref, dt) if solver == "euler" else (
                O.rk4_step(f, 0.0, ref, dt) if solver == "rk4" else O.tsit5_step(f, 0.0, ref, dt)[0])
        assert rel_l2(sol.ys[-1][b] - y0[b], ref - y0[b]) < 1e-10, (solver, b)
        assert abs(sol.ys[-1][b].mean() - y0[b].mean()) < 1e-14  # flux form conserves the mean


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("closures", ["regsol", "cubic", "legendre"])
def test_closure_classes_of_the_3d_kernels_vs_oracle(dtype, closures):
    """the 3-D kernels are instantiated per closure class (fixed polynomial / logit forms unrolled, run-time walk for the
    rest of the family): each against the numpy oracle on a grid that does not fill the launch's blocks"""
    nx, ny, nz = 18, 9, 70
    dom = _dom(nx, ny, nz)
    hx, hy, hz = dom.dx
    if closures == "legendre":
        mu = P.ChemicalPotentialLegendrePolynomials(np.array([0.0, 0.3, -0.1, 0.05, 0.02, -0.01]), prior_fn=lambda c: np.log(c / (1.0 - c)))
        mob = MOB["c1mc"]
    elif closures == "cubic":
        mu, mob = MU["cubic"], (lambda c: 1.0 + c * c)
    else:
        mu, mob = MU["regsol"], MOB["c1mc"]
    eq = P.CahnHilliard3DPeriodic(dom, 0.002, mu, mob)
    rng = np.random.default_rng(3)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((2, nx, ny, nz)), 0.05, 0.95).astype(dtype)
    sol = P.diffeqsolve(eq, P.RK4(), 0.0, 3e-7, 1e-7, y0)
    want_tag = {"regsol": "CH-3D,logit", "cubic": "CH-3D,poly", "legendre": "CH-3D>"}[closures]
    assert want_tag in sol.stats["kernel"], sol.stats["kernel"]
    f = lambda t, u: O.ch3d_rhs_fd(u, hx, hy, hz, 0.002, mu, mob)
    for b in range(2):
        ref = y0[b].astype(np.float64)
        for i in range(3):
            ref = O.rk4_step(f, 0.0, ref, 1e-7)
        got = sol.ys[-1][b].astype(np.float64)
        if dtype is np.float64:
            assert rel_l2(got - y0[b], ref - y0[b]) < 1e-10, (closures, b)
        else:  # the fp32 state's own rounding (6e-8) is the floor of an increment this small
            assert np.max(np.abs(got - ref)) < 1e-6, (closures, b, float(np.max(np.abs(got - ref))))


def test_long_run_replays_the_substep_graph():
    """>= 32 substeps of a small grid replay a captured hipGraph (two kernels per stage in 3-D)"""
    rng = np.random.default_rng(6)
    nx, ny, nz = 16, 16, 24
    dom = _dom(nx, ny, nz)
    hx, hy, hz = dom.dx
    eq = P.CahnHilliard3DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny, nz)), 0.05, 0.95)
    f = lambda t, u: O.ch3d_rhs_fd(u, hx, hy, hz, 0.002, MU["regsol"], MOB["c1mc"])
    dt, n = 1e-7, 64  # four replays of the 16-substep graph, no eager remainder
    sol = P.diffeqsolve(eq, P.RK4(), 0.0, n * dt, dt, y0)
    assert "hipGraph" in sol.stats["kernel"], sol.stats
    ref = y0
    for i in range(n):
        ref = O.rk4_step(f, 0.0, ref, dt)
    assert rel_l2(sol.ys[-1] - y0, ref - y0) < 1e-10


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_imex_3d_vs_oracle_and_pde_model(dtype):
    """docs/notebooks/optimization_3D.ipynb set-up: PDEModel(CahnHilliard3DPeriodic, SemiImplicitFourierSpectral)"""
    rng = np.random.default_rng(9)
    n = 32
    dom = P.Domain((n, n, n), ((-0.16, 0.16),) * 3, "dimensionless")
    params = {"kappa": 0.002, "mu": MU["regsol"], "D": lambda c: 0.15 * np.ones_like(c)}
    y0 = np.clip(0.01 * rng.standard_normal((n, n, n)) + 0.5, 0.0, 1.0).astype(dtype)
    model = P.PDEModel(equation_type=P.CahnHilliard3DPeriodic, domain=dom, solver_type=P.SemiImplicitFourierSpectral)
    ts = np.linspace(0.0, 2e-5, 3)
    sol = model.solve(params, y0, ts, {"A": 0.5}, dt0=0.000001, max_steps=1000000)
    assert sol.shape == (3, n, n, n) and sol.dtype == dtype
    eq = P.CahnHilliard3DPeriodic(dom, **params)
    hx, hy, hz = dom.dx
    rhs = lambda t, u: O.ch3d_rhs_fd(u, hx, hy, hz, 0.002, MU["regsol"], params["D"])
    ref = y0.astype(np.float64)
    for i in range(20):
        ref = O.imex_step(rhs, i * 1e-6, ref, 1e-6, 0.5, eq.fourier_symbol)
    tol = 1e-9 if dtype is np.float64 else 5e-4
    assert rel_l2(sol[-1].astype(np.float64) - y0, ref - y0) < tol
    assert abs(sol[-1].astype(np.float64).mean() - y0.astype(np.float64).mean()) < (1e-14 if dtype is np.float64 else 2e-7)


def test_rhs_fourier_against_reference_goldens(golden):
    """CahnHilliard3DPeriodic.rhs_fourier (cahn_hilliard.py:167-175; 9 transforms) against goldens the
    reference's own source produced (fp64: spectral constants are f64 in numpy)"""
    z = golden("ch3d_fourier.npz")
    tags = sorted(k[: -len("/rhs")] for k in z.files if k.endswith("/rhs"))
    assert len(tags) == 3
    for tag in tags:
        nx, ny, nz = (int(v) for v in tag.split("_")[0].split("x"))
        u, want = z[tag + "/u"], z[tag + "/rhs"]
        eq = P.CahnHilliard3DPeriodic(_dom(nx, ny, nz), 0.002, MU["regsol"], MOB["c1mc"], derivs="fourier")
        got = eq.rhs(u, 0.0)
        assert got.dtype == want.dtype and got.shape == want.shape
        assert rel_l2(got, want) < 1e-11, (tag, rel_l2(got, want))
        assert "rhs_fourier<CH-3D>" in P.engine.default_engine().last_kernel
        # fp32 state: against the same golden at fp32 accuracy of a 9-transform chain with k^4 amplification
        got32 = eq.rhs(u.astype(np.float32), 0.0)
        assert got32.dtype == np.float32 and rel_l2(got32, want) < 5e-4, rel_l2(got32, want)
    # a batch equals its members, and the spectral RHS drives the explicit integrators
    nx, ny, nz = 16, 12, 20
    dom = _dom(nx, ny, nz)
    eq = P.CahnHilliard3DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"], derivs="fourier")
    u = z["16x12x20_float64/u"]
    ub = np.stack([u, np.clip(0.5 * u + 0.2, 0.05, 0.95)])
    np.testing.assert_allclose(eq.rhs(ub, 0.0)[0], eq.rhs(u, 0.0), rtol=0, atol=1e-9 * np.abs(want).max())
    hx, hy, hz = dom.dx
    f = lambda t, v: O.ch3d_rhs_fourier(v, hx, hy, hz, 0.002, MU["regsol"], MOB["c1mc"])
    sol = P.diffeqsolve(eq, P.RK4(), 0.0, 3e-8, 1e-8, ub)
    for b in range(2):
        ref = ub[b]
        for i in range(3):
            ref = O.rk4_step(f, 0.0, ref, 1e-8)
        assert rel_l2(sol.ys[-1][b] - ub[b], ref - ub[b]) < 1e-9


def test_errors():
    dom = _dom(8, 8, 8)
    with pytest.raises(ValueError, match="Invalid derivative type"):
        P.CahnHilliard3DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"], derivs="spectral")
    with pytest.raises(ValueError, match="3-D"):
        P.CahnHilliard3DPeriodic(P.Domain((8, 8), ((0, 1), (0, 1)), "d"), 0.002, MU["regsol"], MOB["c1mc"])
    eq = P.CahnHilliard3DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    with pytest.raises(ValueError, match="does not match"):
        eq.rhs(np.zeros((8, 8)), 0.0)
