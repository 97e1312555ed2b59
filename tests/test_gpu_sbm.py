"""Smoothed-boundary Allen-Cahn / Cahn-Hilliard on the MI355X against the
reference's goldens (tests/golden/sbm_cases.npz, written by oracle/gen_golden.py from
allen_cahn.py:139-156 / cahn_hilliard.py:257-289) and the numpy oracle."""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from pde_opt_amd.numerics.shapes import _ShapeSmoothing
from util import MOB, MU, SBM_F, SBM_FLUX, SBM_THETA, TOL, rel_l2, sbm_domain, sbm_psi

pytestmark = pytest.mark.gpu


def _eq(kind, dom, theta=SBM_THETA, flux=SBM_FLUX, kappa=1.5):
    if kind == "ac":
        return P.AllenCahn2DSmoothedBoundary(dom, kappa, SBM_F, MU["regsol"], MOB["c1mc"], theta)
    return P.CahnHilliard2DSmoothedBoundary(dom, kappa, SBM_F, MU["regsol"], MOB["c1mc"], theta, flux)


def test_rhs_against_reference_goldens(golden):
    z = golden("sbm_cases.npz")
    keys = sorted(k[: -len("/psi")] for k in z.files if k.endswith("/psi"))
    assert len(keys) == 8
    for key in keys:
        kind = key.split("/")[0]
        psi, u = z[key + "/psi"], z[key + "/u"]
        eq = _eq(kind, sbm_domain(P, psi.astype(np.float64)))
        for t in (0.0, 0.17):
            want = z[f"{key}/rhs_t{t}"]
            got = eq.rhs(u, t)
            # (the fp32 goldens came out as float64: numpy promotes float32 * np.cos(float) where JAX
            # would stay in fp32; the inputs are fp32 and so is the kernel)
            assert got.dtype == u.dtype
            assert rel_l2(got, want) < TOL[u.dtype], (key, t, rel_l2(got, want))
    assert "SBM" in P.engine.default_engine().last_kernel


def _oracle_rhs(kind, psi, lh):
    if kind == "ac":
        return lambda t, u: O.ac_sbm_rhs(u, psi, 1.0, 1.0, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA(t), lh)
    return lambda t, u: O.ch_sbm_rhs(u, psi, 1.0, 1.0, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA(t),
                                     SBM_FLUX(t), lh)


@pytest.mark.parametrize("kind", ["ac", "ch"])
@pytest.mark.parametrize("solver", ["euler", "rk4", "tsit5"])
def test_trajectory_with_time_dependent_contact_angle(kind, solver):
    """theta(t) / flux(t) are evaluated at every stage time (t, t+dt/2, t+dt; Tsit5's c_i)."""
    rng = np.random.default_rng(11)
    psi = sbm_psi(72, 120) if kind == "ac" else sbm_psi(96, 40)
    eq = _eq(kind, sbm_domain(P, psi))
    y0 = np.clip(0.5 + 0.1 * rng.standard_normal((2,) + psi.shape), 0.1, 0.9)
    f = _oracle_rhs(kind, psi, eq.left_half)
    dt, n, t0 = (2e-3 if kind == "ch" else 2e-2), 4, 0.03
    s = {"euler": P.Euler(), "rk4": P.RK4(), "tsit5": P.Tsit5()}[solver]
    sol = P.diffeqsolve(eq, s, t0=t0, t1=t0 + n * dt, dt0=dt, y0=y0)
    for b in range(2):
        ref = y0[b]
        for i in range(n):
            t = t0 + i * dt
            if solver == "euler":
                ref = O.euler_step(f, t, ref, dt)
            elif solver == "rk4":
                ref = O.rk4_step(f, t, ref, dt)
            else:
                ref = O.tsit5_step(f, t, ref, dt)[0]
        assert rel_l2(sol.ys[-1][b] - y0[b], ref - y0[b]) < 1e-10, (kind, solver)


def test_psi_weighted_mass_balance_and_adaptive_run():
    """sum(psi u) changes only through the boundary flux: d/dt sum(psi u) = sum(|grad psi|) flux(t)
    (cahn_hilliard.py:282-289 multiplied by psi).  Run adaptively (Tsit5 + PID) like
    notebooks/smooth_boundary.ipynb, at a size beyond the oracle's reach in test time."""
    n = 384
    psi = sbm_psi(n, n)
    dom = sbm_domain(P, psi)
    rng = np.random.default_rng(2)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((n, n)), 0.1, 0.9)
    flux = lambda t: 0.01  # noqa: E731
    eq = _eq("ch", dom, flux=flux)
    sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.05, 1e-4, y0, stepsize_controller=P.PIDController(rtol=1e-5, atol=1e-7))
    assert sol.stats["num_accepted_steps"] >= 3
    m0, m1 = np.sum(psi * y0), np.sum(psi * sol.ys[-1])
    want = 0.05 * 0.01 * np.sum(eq.norm_grad_psi * psi)
    assert abs((m1 - m0) - want) < 1e-9 * abs(m0), (m1 - m0, want)
    # without flux the psi-weighted mass is conserved to rounding
    eq0 = _eq("ch", dom, flux=lambda t: 0.0)
    sol0 = P.diffeqsolve(eq0, P.RK4(), 0.0, 5e-3, 1e-3, y0)
    assert abs(np.sum(psi * sol0.ys[-1]) - m0) < 1e-11 * abs(m0)


def test_fp32_and_batch_consistency():
    rng = np.random.default_rng(8)
    psi = sbm_psi(64, 128)
    eq = _eq("ac", sbm_domain(P, psi))
    y0 = np.clip(0.5 + 0.1 * rng.standard_normal((3,) + psi.shape), 0.1, 0.9).astype(np.float32)
    batch = P.diffeqsolve(eq, P.RK4(), 0.0, 0.08, 0.02, y0).ys[-1]
    for b in range(3):
        single = P.diffeqsolve(eq, P.RK4(), 0.0, 0.08, 0.02, y0[b]).ys[-1]
        np.testing.assert_array_equal(single, batch[b])
    f = _oracle_rhs("ac", psi, eq.left_half)
    ref = y0[0].astype(np.float64)
    for i in range(4):
        ref = O.rk4_step(f, i * 0.02, ref, 0.02)
    assert rel_l2(batch[0] - y0[0], ref - y0[0]) < 5e-4


def test_missing_aux_is_an_error():
    eng = P.HipEngine()
    from pde_opt_amd import _lib as L
    from pde_opt_amd.numerics.closures import as_closure

    eng.configure(L.EQ_ALLEN_CAHN_SBM, np.float64, 16, 16, 1, 1.0, 1.0, 1.0, as_closure(MU["regsol"]),
                  as_closure(MOB["c1mc"]), fe=as_closure(SBM_F))
    eng.set_state(np.full((16, 16), 0.5))
    with pytest.raises(P.PdeoptError, match="SBM_PSI"):
        eng.rhs(0.0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_two_pass_ch_sbm_equals_the_literal_one_pass_kernel(dtype):
    """CH-SBM evaluates `inner` once per cell into a work field and then the flux divergence (two launches); the
    literal kernel re-evaluates it at 5 points per cell.  Same expressions, equal to rounding, batched and per stage."""
    from pde_opt_amd import _lib as L

    rng = np.random.default_rng(21)
    psi = sbm_psi(80, 72)
    eq = _eq("ch", sbm_domain(P, psi))
    y0 = np.clip(0.5 + 0.1 * rng.standard_normal((3,) + psi.shape), 0.1, 0.9).astype(dtype)
    outs, kernels = [], []
    for literal in (False, True):
        eng = P.HipEngine()
        eng.set_kernel_path(L.PATH_GENERIC)        # the one-thread-per-cell forms (the LDS-tiled kernel is the default)
        eng.set_fuse_stages(-1 if literal else 1)  # auto picks by size: two passes from 2^18 cells on
        sol = P.diffeqsolve(eq, P.RK4(), 0.02, 0.02 + 5 * 2e-3, 2e-3, y0, engine=eng)
        outs.append(sol.ys[-1])
        kernels.append(eng.last_kernel)
        eng.close()
    assert kernels == ["stage_two_pass<CH-SBM>", "stage_generic<CH-SBM>"], kernels
    assert np.isfinite(outs[0]).all() and np.any(outs[0] != y0)
    # the two kernels are different instantiations: hipcc contracts their FMAs differently, a few ulp of the state
    inc0, inc1 = outs[0].astype(np.float64) - y0, outs[1].astype(np.float64) - y0
    assert rel_l2(inc0, inc1) < (1e-12 if dtype is np.float64 else 2e-5), rel_l2(inc0, inc1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["ac", "ch"])
@pytest.mark.parametrize("shape", [(64, 128), (100, 100), (80, 72), (8, 16), (144, 264)])
def test_lds_tiled_sbm_kernel_vs_oracle_and_generic(kind, shape, dtype):
    """csrc/stencil_sbm_tiled.hpp: u and psi tiles in LDS, `inner` once per point, every integrator
    through it (Euler, RK4, Tsit5's fused next-stage form) -- against the numpy oracle and the one-thread-per-cell
    kernels it restates term for term; tile-divisible and ragged grids, a batch of 3"""
    from pde_opt_amd import _lib as L

    rng = np.random.default_rng(31)
    psi = sbm_psi(*shape)
    eq = _eq(kind, sbm_domain(P, psi))
    y0 = np.clip(0.5 + 0.1 * rng.standard_normal((3,) + psi.shape), 0.1, 0.9).astype(dtype)
    f = _oracle_rhs(kind, psi, eq.left_half)
    dt, n, t0 = (2e-3 if kind == "ch" else 2e-2), 3, 0.03
    for solver, step in ((P.RK4(), O.rk4_step), (P.Euler(), O.euler_step), (P.Tsit5(), lambda ff, t, y, h: O.tsit5_step(ff, t, y, h)[0])):
        outs = {}
        for path in (L.PATH_AUTO, L.PATH_GENERIC):
            eng = P.HipEngine()
            eng.set_kernel_path(path)
            outs[path] = P.diffeqsolve(eq, solver, t0, t0 + n * dt, dt, y0, engine=eng).ys[-1]
            assert ("sbm_tiled" in eng.last_kernel) == (path == L.PATH_AUTO), eng.last_kernel
            eng.close()
        ref = y0[1].astype(np.float64)
        for i in range(n):
            ref = step(f, t0 + i * dt, ref, dt)
        tol = 1e-10 if dtype is np.float64 else 5e-4
        assert rel_l2(outs[L.PATH_AUTO][1] - y0[1], ref - y0[1]) < tol, (kind, shape, type(solver).__name__)
        d = rel_l2(outs[L.PATH_AUTO].astype(np.float64) - y0, outs[L.PATH_GENERIC].astype(np.float64) - y0)
        assert d < (1e-12 if dtype is np.float64 else 2e-5), (kind, shape, d)
    # the right-hand side alone, against the oracle
    got = eq.rhs(y0, 0.17)
    assert "sbm_tiled" in P.engine.default_engine().last_kernel
    for b in range(3):
        assert rel_l2(got[b], f(0.17, y0[b].astype(np.float64))) < TOL[np.dtype(dtype)]


def test_time_terms_are_sampled_once_per_advance_not_per_stage():
    """fixed-step Euler / RK4: theta(t), flux(t) at every stage time of the call go to the library as ONE table
    (pdeopt_set_time_table) -- the substep loop makes no host callback; the result equals the per-stage callback's"""
    from pde_opt_amd import _lib as L

    rng = np.random.default_rng(5)
    psi = sbm_psi(64, 128)
    calls = []
    theta = lambda t: (calls.append(t), SBM_THETA(t))[1]  # noqa: E731
    eq = _eq("ch", sbm_domain(P, psi), theta=theta)
    y0 = np.clip(0.5 + 0.1 * rng.standard_normal(psi.shape), 0.1, 0.9)
    eng = P.HipEngine()
    eng.set_small_persist(-1)  # the tiled kernel with host-sampled time terms (a polynomial theta(t) would otherwise be evaluated in-kernel)
    sol = P.diffeqsolve(eq, P.RK4(), 0.03, 0.03 + 50 * 2e-3, 2e-3, y0, engine=eng)
    # (the upload also probes theta for being a polynomial in t -- closures.poly_in_t: a symbol and the times 0, 0.37, 1.9,
    # outside this run's stage times -- for the in-kernel adaptive solve; not counted here)
    stage_calls = lambda: [t for t in calls if isinstance(t, float) and 0.03 <= t <= 0.1301]  # noqa: E731
    n_table = len(stage_calls())
    assert 0 < n_table <= 151  # 50 substeps x {t, t + dt/2, t + dt}, duplicates merged -- sampled before the launch loop
    calls.clear()
    # the callback path (table cleared by hand): 4 calls per substep, same bits
    eng2 = P.HipEngine()
    eng2.set_small_persist(-1)
    eng2._upload_time_table = lambda *a, **k: None
    sol2 = P.diffeqsolve(eq, P.RK4(), 0.03, 0.03 + 50 * 2e-3, 2e-3, y0, engine=eng2)
    assert len(stage_calls()) == 200
    np.testing.assert_array_equal(sol.ys[-1], sol2.ys[-1])
    eng.close()
    eng2.close()


# ----------------------------------------------------------------------------- Shape (shapes.py:21-79)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_shape_smoothing_rhs_against_reference_goldens(golden, dtype):
    """the PDEOPT_EQ_SHAPE_SMOOTH kernel against what the reference's own ``rhs`` closure of
    ``Shape.smooth_shape`` (shapes.py:44-64) returned for the same fields (tests/golden/shapes.npz)"""
    z = golden("shapes.npz")
    keys = sorted(k[: -len("/rhs")] for k in z.files if k.endswith("/rhs"))
    assert len(keys) == 6
    for key in keys:
        par = key.split("/")[1].split("_")
        dx, eps, c = (float(par[0][2:]), float(par[1])), float(par[2][3:]), float(par[3][1:])
        u = z[key + "/u"].astype(dtype)
        nx, ny = u.shape
        eq = _ShapeSmoothing(P.Domain((nx, ny), ((0.0, nx * dx[0]), (0.0, ny * dx[1])), "dimensionless"), eps, c)
        got = eq.rhs(u, 0.0)
        assert got.dtype == dtype
        want = z[key + "/rhs"] if dtype is np.float64 else O.shape_smooth_rhs(u.astype(np.float64), *dx, eps, c)
        assert rel_l2(got, want) < (1e-12 if dtype is np.float64 else 2e-5), (key, rel_l2(got, want))
    assert "shape-smooth" in P.engine.default_engine().last_kernel


def test_shape_smooth_field_and_smoothed_boundary_equation_on_it(golden):
    """``Shape(mask).smooth`` from the GPU solve equals the same adaptive driver run on the oracle, and feeds
    ``Domain(geometry=shape)`` -> ``AllenCahn2DSmoothedBoundary`` like the reference's notebook does"""
    from fake_engine import OracleEngine

    z = golden("shapes.npz")
    mask = z["disc48x40/mask"]
    shape = P.Shape(mask, smooth_epsilon=1.5, smooth_curvature=0.2, smooth_tf=0.8)
    ref = P.Shape(mask, smooth_epsilon=1.5, smooth_curvature=0.2, smooth_tf=0.8, engine=OracleEngine())
    np.testing.assert_allclose(shape.smooth, ref.smooth, rtol=0, atol=1e-9)
    assert shape.smooth.min() == 0.001 and shape.smooth.max() == 1.0
    assert np.mean((shape.smooth > 0.01) & (shape.smooth < 0.98)) > 0.05  # a diffuse interface exists
    nx, ny = mask.shape
    dom = P.Domain((nx, ny), ((0.0, float(nx)), (0.0, float(ny))), "dimensionless", geometry=shape)
    eq = _eq("ac", dom)
    u = np.clip(0.5 + 0.1 * np.random.default_rng(0).standard_normal((nx, ny)), 0.1, 0.9)
    got = eq.rhs(u, 0.1)
    want = O.ac_sbm_rhs(u, shape.smooth, 1.0, 1.0, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA(0.1), eq.left_half)
    assert rel_l2(got, want) < 1e-11
