"""Host-side logic of the product on the CPU: the diffeqsolve driver (step plans, SaveAt linear dense
output, the PID step-size controller) and the PDEEnv protocol, run on an oracle-backed engine double."""
import numpy as np
import pytest

import pde_opt_amd as P
from fake_engine import OracleEngine
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import MOB, MU, std_domain


def _ac(n=24, m=20):
    dom = std_domain(P, n, m)
    return dom, P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])


def test_constant_step_driver_issues_one_advance_per_segment():
    dom, eq = _ac()
    y0 = 0.1 * np.random.default_rng(0).standard_normal(dom.points)
    eng = OracleEngine()
    sol = P.diffeqsolve(eq, P.Euler(), 0.0, 3.5e-4, 1e-4, y0, engine=eng)
    # 3 full steps in ONE library call + the clipped remainder
    adv = [c for c in eng.calls if c[0] == "advance"]
    assert [(c[3], round(c[2], 12)) for c in adv] == [(3, 1e-4), (1, 0.5e-4)]
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
    want = O.integrate(lambda t, y, dt: O.euler_step(f, t, y, dt), y0, 0.0, 3.5e-4, 1e-4)
    np.testing.assert_allclose(sol.ys[-1], want, rtol=0, atol=1e-15)
    assert sol.stats["num_steps"] == 4


def test_saveat_interpolation_matches_oracle_driver():
    dom, eq = _ac()
    y0 = 0.1 * np.random.default_rng(1).standard_normal(dom.points)
    ts = [0.0, 0.4e-4, 1e-4, 2.5e-4, 3.3e-4, 5e-4]
    sol = P.diffeqsolve(eq, P.RK4(), ts[0], ts[-1], 1e-4, y0, saveat=P.SaveAt(ts=ts), engine=OracleEngine())
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
    want = O.solve_saveat(lambda t, y, dt: O.rk4_step(f, t, y, dt), y0, ts, 1e-4)
    np.testing.assert_allclose(sol.ys, want, rtol=0, atol=1e-15)
    np.testing.assert_allclose(sol.ts, ts)


def test_pid_controller_reaches_tanh_profile():
    """tests/test_solvers.py:64-104 on a coarse grid: Tsit5 + PIDController(rtol 1e-4, atol 1e-6)"""
    nx = 64
    dom = std_domain(P, nx, 1)
    kappa = 0.002
    eq = P.AllenCahn2DPeriodic(dom, kappa, lambda c: c**3 - c, lambda c: np.ones_like(c))
    u0 = np.ones((nx, 1))
    u0[: nx // 2] = -1.0
    sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, 2.0, 5e-5, u0, saveat=P.SaveAt(ts=[0.0, 1.0, 2.0]),
                        stepsize_controller=P.PIDController(rtol=1e-4, atol=1e-6), engine=OracleEngine())
    assert sol.ys.shape == (3, nx, 1)
    x = dom.axes()[0]
    analytic = np.tanh(x / np.sqrt(2 * kappa))
    mid = slice(nx // 4, 3 * nx // 4)
    np.testing.assert_allclose(sol.ys[-1].squeeze()[mid], analytic[mid], rtol=1e-2, atol=1e-2)
    st = sol.stats
    assert st["num_accepted_steps"] > 20 and st["num_rejected_steps"] >= 0
    # the controller grew the step far beyond dt0 (stiffness-limited, not stuck at 5e-5)
    assert 2.0 / st["num_accepted_steps"] > 20 * 5e-5
    with pytest.raises(ValueError, match="embedded pair"):
        P.diffeqsolve(eq, P.RK4(), 0.0, 1.0, 1e-3, u0, stepsize_controller=P.PIDController(1e-3, 1e-6), engine=OracleEngine())


def test_imex_driver_uploads_symbol_and_matches_oracle():
    dom = std_domain(P, 16, 12)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
    y0 = 0.1 * np.random.default_rng(2).standard_normal(dom.points)
    eng = OracleEngine()
    sol = P.diffeqsolve(eq, solver, 0.0, 5e-5, 1e-5, y0, engine=eng)
    hx, hy = dom.dx
    sym = O.ch_fourier_symbol(16, 12, hx, hy, 0.002)
    np.testing.assert_array_equal(eng.symbol, sym)
    rhs = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
    want = O.integrate(lambda t, u, dt: O.imex_step(rhs, t, u, dt, 0.5, sym), y0, 0.0, 5e-5, 1e-5)
    np.testing.assert_allclose(sol.ys[-1], want, rtol=0, atol=1e-15)


def test_input_validation():
    dom, eq = _ac()
    with pytest.raises(ValueError, match="does not match domain"):
        P.diffeqsolve(eq, P.Euler(), 0.0, 1e-3, 1e-4, np.zeros((5, 5)), engine=OracleEngine())
    with pytest.raises(ValueError, match="complex states"):
        P.diffeqsolve(eq, P.Euler(), 0.0, 1e-3, 1e-4, np.zeros(dom.points, complex), engine=OracleEngine())
    with pytest.raises(ValueError, match="positive"):
        P.diffeqsolve(eq, P.Euler(), 0.0, 1e-3, 0.0, np.zeros(dom.points), engine=OracleEngine())


def test_smoothed_boundary_host_wiring():
    """SBM equations: closures traced (incl. the mixing-entropy free energy), psi-derived fields
    uploaded, theta(t)/flux(t) delivered per RHS evaluation time; Tsit5 stage times reach them."""
    from util import SBM_F, SBM_FLUX, SBM_THETA, sbm_domain, sbm_psi

    psi = sbm_psi(60, 28)
    dom = sbm_domain(P, psi)
    eq = P.CahnHilliard2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA, SBM_FLUX)
    assert eq._f_desc.flags == 4 and eq._f_desc.coef == (0.059, 3.0, -3.0)
    assert eq.left_half[:50].all() and not eq.left_half[50:].any()
    np.testing.assert_array_equal(eq.norm_grad_psi, O.sbm_norm_grad(psi, 1.0, 1.0))
    y0 = np.clip(0.5 + 0.1 * np.random.default_rng(5).standard_normal(psi.shape), 0.1, 0.9)
    f = lambda t, u: O.ch_sbm_rhs(u, psi, 1.0, 1.0, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA(t),
                                  SBM_FLUX(t), eq.left_half)
    dt = 2e-3
    sol = P.diffeqsolve(eq, P.RK4(), 0.05, 0.05 + 3 * dt, dt, y0, engine=OracleEngine())
    want = y0
    for i in range(3):
        want = O.rk4_step(f, 0.05 + i * dt, want, dt)
    np.testing.assert_allclose(sol.ys[-1], want, rtol=0, atol=1e-14)

    ac = P.AllenCahn2DSmoothedBoundary(sbm_domain(P, sbm_psi(20, 120)), 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA)
    assert ac.left_half[:, :100].all() and not ac.left_half[:, 100:].any()
    assert ac._time_terms(0.1) == (np.cos(SBM_THETA(0.1)), 0.0, 0.0)
    with pytest.raises(ValueError, match="Invalid derivative type"):
        P.AllenCahn2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA, derivs="fourier")
    with pytest.raises(ValueError, match="geometry"):
        P.AllenCahn2DSmoothedBoundary(std_domain(P, 8, 8), 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA)
