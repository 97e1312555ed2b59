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


def test_tsit5_dense_output_weights_and_order():
    """The 4th-order continuous extension of Tsitouras' pair (the interpolant diffrax.Tsit5 evaluates at
    SaveAt points inside a step): b_i(1) are the 5th-order weights, b_i(0) = 0, the quadrature conditions
    sum b_i c_i^q = theta^(q+1)/(q+1) hold for q = 0..3, and the local error at mid-step falls as h^5."""
    C = np.array((0.0,) + O._TS_C)
    np.testing.assert_allclose(O.tsit5_dense_weights(1.0), O._TS_B, rtol=0, atol=2e-15)
    np.testing.assert_array_equal(O.tsit5_dense_weights(0.0), np.zeros(7))
    for th in (0.1, 0.37, 0.5, 0.93):
        b = np.array(O.tsit5_dense_weights(th))
        for q in range(4):
            assert abs((b * C**q).sum() - th ** (q + 1) / (q + 1)) < 1e-14
    f = lambda t, y: -y + np.sin(3 * t)
    exact = lambda t: 0.1 * (np.sin(3 * t) - 3 * np.cos(3 * t)) + 1.3 * np.exp(-t)  # y' = -y + sin 3t, y(0) = 1
    errs = []
    for h in (0.2, 0.1, 0.05):
        y0 = np.array([exact(0.0)])
        _, _, _, ks = O.tsit5_step(f, 0.0, y0, h, return_slopes=True)
        errs.append(abs(O.tsit5_dense(y0, h, ks, 0.5)[0] - exact(0.5 * h)))
    assert errs[0] / errs[1] > 24 and errs[1] / errs[2] > 24, errs  # ~2^5


def test_saveat_several_points_inside_one_step():
    """two save points inside the same step, then one in the next: the step is taken once (RK4: linear dense
    output; Tsit5: its 4th-order interpolant, as diffrax)"""
    dom, eq = _ac()
    y0 = 0.1 * np.random.default_rng(4).standard_normal(dom.points)
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
    dt = 1e-4
    ts = [0.0, 0.2e-4, 0.7e-4, 1.6e-4, 1.9e-4, 2e-4, 3e-4]
    eng = OracleEngine()
    sol = P.diffeqsolve(eq, P.RK4(), ts[0], ts[-1], dt, y0, saveat=P.SaveAt(ts=ts), engine=eng)
    want = O.solve_saveat(lambda t, y, h: O.rk4_step(f, t, y, h), y0, ts, dt)
    np.testing.assert_allclose(sol.ys, want, rtol=0, atol=1e-15)
    assert sum(c[3] for c in eng.calls if c[0] == "advance") == 3  # three steps in total, none repeated
    sol = P.diffeqsolve(eq, P.Tsit5(), ts[0], ts[-1], dt, y0, saveat=P.SaveAt(ts=ts), engine=OracleEngine())
    y, out = y0, [y0]
    for i, inner in enumerate(([0.2, 0.7], [0.6, 0.9], [])):
        y1, _, _, ks = O.tsit5_step(f, i * dt, y, dt, return_slopes=True)
        out += [O.tsit5_dense(y, dt, ks, th) for th in inner]
        y = y1
        if i >= 1:
            out.append(y)
    np.testing.assert_allclose(sol.ys, np.stack(out), rtol=0, atol=1e-15)


def test_adaptive_tsit5_interior_save_points_are_fourth_order():
    """tests/test_solvers.py:64-104 saves 200 interior points under Tsit5 + PIDController; here the interior
    points of a coarse adaptive solve are compared with a fine fixed-step solve: the 4th-order dense output
    keeps them at the accuracy of the steps themselves (linear interpolation would be off by O(h^2))."""
    nx = 48
    dom = std_domain(P, nx, 1)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
    u0 = np.ones((nx, 1))
    u0[: nx // 2] = -1.0
    ts = np.linspace(0.0, 0.5, 41)
    sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.5, 5e-5, u0, saveat=P.SaveAt(ts=ts),
                        stepsize_controller=P.PIDController(rtol=1e-6, atol=1e-8), engine=OracleEngine())
    assert sol.stats["num_accepted_steps"] < 4 * len(ts)  # several save points per step do occur
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
    ref, y, k = [u0], u0, 0
    for i in range(1, 2001):  # fine reference: 2000 RK4 steps of 2.5e-4
        y = O.rk4_step(f, 0.0, y, 2.5e-4)
        if i % 50 == 0:
            ref.append(y)
    err = np.max(np.abs(sol.ys - np.stack(ref)))
    assert err < 2e-5, err


def test_per_environment_step_sizes_equal_solo_solves():
    """PIDController(per_environment=True): each environment of a batch takes the steps it would take alone
    (per-env error norm AND per-env dt); the default shares the worst-case step."""
    nx = 48
    dom = std_domain(P, nx, 1)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
    u = np.ones((nx, 1))
    u[: nx // 2] = -1.0
    rng = np.random.default_rng(9)
    y0 = np.stack([u, 0.05 * rng.standard_normal((nx, 1)), 0.9 * u + 0.3 * rng.standard_normal((nx, 1))])
    ts = [0.0, 0.013, 0.05, 0.2]
    ctl = dict(rtol=1e-5, atol=1e-7)
    solo = [P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0[b], saveat=P.SaveAt(ts=ts),
                          stepsize_controller=P.PIDController(**ctl), engine=OracleEngine()) for b in range(3)]
    both = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(ts=ts),
                         stepsize_controller=P.PIDController(**ctl, per_environment=True), engine=OracleEngine())
    assert both.ys.shape == (4, 3, nx, 1)
    for b in range(3):
        np.testing.assert_allclose(both.ys[:, b], solo[b].ys, rtol=0, atol=1e-12)
        assert both.stats["num_accepted_steps"][b] == solo[b].stats["num_accepted_steps"]
        assert both.stats["num_rejected_steps"][b] == solo[b].stats["num_rejected_steps"]
    assert len(set(both.stats["num_accepted_steps"])) > 1  # the environments really do step differently
    shared = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(ts=ts),
                           stepsize_controller=P.PIDController(**ctl), engine=OracleEngine())
    assert shared.stats["num_accepted_steps"] >= max(both.stats["num_accepted_steps"])


def _engine_in_kernel():
    e = OracleEngine()
    e.small_adaptive = True  # diffeqsolve routes to integrate._solve_adaptive_in_kernel (the one-launch solve's driver)
    return e


@pytest.mark.parametrize("saveat", [dict(t1=True), dict(ts=[0.0, 0.013, 0.05, 0.2]), dict(t0=True, ts=[0.02, 0.2], t1=True),
                                    dict(ts=[-1.0, 0.0, 0.1])])
def test_one_launch_adaptive_driver_equals_the_step_by_step_loop(saveat):
    """integrate._solve_adaptive_in_kernel (what drives pdeopt_tsit5_solve_small) against integrate._solve_adaptive on
    the same oracle arithmetic: save times, their order, the states, the statistics -- one environment and several
    with their own controllers"""
    nx = 48
    dom = std_domain(P, nx, 1)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
    rng = np.random.default_rng(10)
    y0 = np.stack([np.where(np.arange(nx)[:, None] < nx // 2, -1.0, 1.0), 0.05 * rng.standard_normal((nx, 1)),
                   0.5 * rng.standard_normal((nx, 1))])
    ctl = dict(rtol=1e-5, atol=1e-7, pcoeff=0.2, icoeff=0.6)
    a = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0[0], saveat=P.SaveAt(**saveat), stepsize_controller=P.PIDController(**ctl),
                      engine=_engine_in_kernel())
    b = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0[0], saveat=P.SaveAt(**saveat), stepsize_controller=P.PIDController(**ctl),
                      engine=OracleEngine())
    np.testing.assert_array_equal(a.ts, b.ts)
    np.testing.assert_array_equal(a.ys, b.ys)
    assert {k: a.stats[k] for k in ("num_steps", "num_accepted_steps", "num_rejected_steps")} == {
        k: b.stats[k] for k in ("num_steps", "num_accepted_steps", "num_rejected_steps")}
    per = P.PIDController(**ctl, per_environment=True)
    a = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(**saveat), stepsize_controller=per, engine=_engine_in_kernel())
    b = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(**saveat), stepsize_controller=per, engine=OracleEngine())
    np.testing.assert_array_equal(a.ts, b.ts)
    np.testing.assert_allclose(a.ys, b.ys, rtol=0, atol=1e-12)  # the batched host loop scales slopes by dt_b / dt_ref
    assert a.stats["num_accepted_steps"] == b.stats["num_accepted_steps"]
    assert a.stats["num_rejected_steps"] == b.stats["num_rejected_steps"]
    # a step size shared by the batch is not the one-launch solve's job
    c = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(**saveat), stepsize_controller=P.PIDController(**ctl),
                      engine=_engine_in_kernel())
    d = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(**saveat), stepsize_controller=P.PIDController(**ctl),
                      engine=OracleEngine())
    np.testing.assert_array_equal(c.ys, d.ys)


def test_one_launch_adaptive_driver_step_budget_and_stall():
    nx = 32
    dom = std_domain(P, nx, 1)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
    y0 = np.where(np.arange(nx)[:, None] < nx // 2, -1.0, 1.0).astype(np.float64)
    ctl = P.PIDController(rtol=1e-6, atol=1e-9)
    ts = list(np.linspace(0.0, 0.2, 6))
    with pytest.raises(RuntimeError, match="max_steps=5"):
        P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(ts=ts), stepsize_controller=ctl, max_steps=5,
                      engine=_engine_in_kernel())
    a = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(ts=ts, t1=True), stepsize_controller=ctl, max_steps=5,
                      throw=False, engine=_engine_in_kernel())
    b = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(ts=ts, t1=True), stepsize_controller=ctl, max_steps=5,
                      throw=False, engine=OracleEngine())
    np.testing.assert_array_equal(a.ts, b.ts)
    np.testing.assert_array_equal(a.ys, b.ys)
    assert a.stats["num_steps"] == b.stats["num_steps"] == 5 and a.ts[-1] < 0.2
    # per-environment: every save time is reported, the slots an environment never reached are NaN
    per = P.PIDController(rtol=1e-6, atol=1e-9, per_environment=True)
    yb = np.stack([y0, 0.1 * y0])
    a = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, yb, saveat=P.SaveAt(ts=ts), stepsize_controller=per, max_steps=5, throw=False,
                      engine=_engine_in_kernel())
    b = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, yb, saveat=P.SaveAt(ts=ts), stepsize_controller=per, max_steps=5, throw=False,
                      engine=OracleEngine())
    np.testing.assert_array_equal(a.ts, b.ts)
    assert np.array_equal(np.isnan(a.ys), np.isnan(b.ys)) and np.isnan(a.ys).any()
    np.testing.assert_allclose(np.nan_to_num(a.ys), np.nan_to_num(b.ys), rtol=0, atol=1e-12)
    # a step that cannot advance t ends the launch instead of spinning; the driver reports it
    with pytest.raises(RuntimeError, match="underflow"):
        P.diffeqsolve(eq, P.Tsit5(), 1e20, 1e20 + 1e5, 1e-4, y0, stepsize_controller=P.PIDController(rtol=1e-6, atol=1e-9, dtmax=1e-3),
                      engine=_engine_in_kernel())


def test_check_convergence_harness_on_the_oracle_engine(monkeypatch):
    """``numerics.utils.testing.check_convergence`` + ``numerics.symbolic`` (the reference's
    tests/test_rhs_convergence.py harness) with the oracle standing in for the GPU: second order for AC and CH."""
    import sympy as sp

    from pde_opt_amd import engine as E
    from pde_opt_amd.numerics.symbolic import SymbolicAllenCahn2DPeriodic, SymbolicCahnHilliard2DPeriodic
    from pde_opt_amd.numerics.utils.testing import check_convergence, convergence_slope, l2_rel_err

    monkeypatch.setitem(E._default_engines, 0, OracleEngine())
    x, y, t = sp.symbols("x y t", real=True)
    u_star = sp.sin(2 * x) * sp.cos(3 * y) * sp.exp(-0.7 * t)
    mu_sym = lambda u: u**3 - u
    R_sym = lambda u: 1 + u**2
    for numeric, symbolic, key in ((P.AllenCahn2DPeriodic, SymbolicAllenCahn2DPeriodic, "R"),
                                   (P.CahnHilliard2DPeriodic, SymbolicCahnHilliard2DPeriodic, "D")):
        dxs, errs = check_convergence(numeric, symbolic, {"kappa": 1e-2, "mu": mu_sym, key: R_sym, "derivs": "fd"},
                                      {"kappa": 1e-2, "mu_sym": mu_sym, key + "_sym": R_sym, "u_star": u_star},
                                      [32, 64, 128], 2 * np.pi)
        assert dxs == [2 * np.pi / n for n in (32, 64, 128)]
        np.testing.assert_allclose(convergence_slope(dxs, errs), 2.0, rtol=0.1)
    assert l2_rel_err([1.0, 2.0], [1.0, 2.0]) == 0.0 and abs(l2_rel_err([3.0, 0.0], [0.0, 4.0]) - 1.25) < 1e-15


def test_shape_host_side_against_reference_goldens(golden):
    """``Shape`` (shapes.py:21-203): the clamps of __post_init__, the arguments of the smoothing solve, the mask's
    graph Laplacian and its lowest modes -- against arrays the reference's own class produced (gen_golden.py)"""
    z = golden("shapes.npz")
    for name in ("disc48x40", "tee32"):
        mask = z[name + "/mask"]
        eng = OracleEngine()
        shape = P.Shape(mask, dx=(0.5, 0.8), smooth_epsilon=2.0, smooth_curvature=0.3, engine=eng)
        key = f"{name}/dx0.5_0.8_eps2.0_c0.3"
        t0, t1, dt0 = z[key + "/solve_args"]
        assert (t0, t1, dt0) == (0.0, shape.smooth_tf, shape.smooth_dt)
        assert eng.eq == L.EQ_SHAPE_SMOOTH and (eng.hx, eng.hy) == (0.5, 0.8)
        assert eng.kappa_env[0] == 0.3 and eng.gpe_k_env[0] == 2.0
        # upstream's clamps (0.001 below, 1 above 0.99) act on whatever the solve returned; on the unsmoothed
        # mask they give the golden, on the smoothed field they bound it
        clamped = np.where(mask < 0.001, 0.001, np.where(mask > 0.99, 1.0, mask))
        np.testing.assert_array_equal(clamped, z[key + "/post_init_of_y0"])
        assert shape.smooth.min() >= 0.001 and shape.smooth.max() <= 1.0
        assert 0.2 < np.mean(np.abs(shape.smooth - mask) > 0.02) < 0.9  # the interface did get smeared
        assert abs(shape.smooth.sum() - mask.sum()) < 0.05 * mask.sum()
        for periodic in (False, True):
            lap, ids = shape.laplacian_from_mask(periodic=periodic)
            np.testing.assert_array_equal(lap.toarray(), z[f"{name}/laplacian_{'periodic' if periodic else 'open'}"])
            np.testing.assert_array_equal(ids, z[name + "/ids"])
        shape.get_shape_modes(6)
        np.testing.assert_allclose(shape.shape_basis_evals, z[name + "/mode_evals"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.abs(shape.shape_basis).sum(axis=(0, 1)), z[name + "/mode_basis_abs_sum"], rtol=1e-6)
    empty = P.Shape(np.zeros((8, 8)), engine=OracleEngine())
    lap, ids = empty.laplacian_from_mask()
    assert lap.shape == (0, 0) and (ids == -1).all()
