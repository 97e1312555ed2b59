"""The multi-workgroup in-kernel adaptive solve (csrc/stencil_coop_adaptive.hpp): Tsit5 + diffrax.PIDController for
the reference's own notebook workloads -- CahnHilliard2DSmoothedBoundary 100^2 (notebooks/smooth_boundary.ipynb:228,397;
cahn_hilliard.py:204-289), advection-diffusion 64^2 (notebooks/run_advection_diffusion.ipynb:84), periodic grids beyond one
compute unit -- several workgroups per environment, one exchange + one barrier per trial step.

The gate is the one of tests/test_gpu_adaptive.py: the same solve driven step by step on the CPU oracle under the
package's own host loop (tests/fake_engine.py: OracleEngine): fp64 takes the SAME accept / reject sequence and agrees to
1e-9 at the save points; fp32 within the controller's tolerance."""
import numpy as np
import pytest

import pde_opt_amd as P
from fake_engine import OracleEngine
from util import MOB, MU, SBM_F, rel_l2, sbm_domain, sbm_psi, std_domain

pytestmark = pytest.mark.gpu


def _solve_pair(eq, y0, t1, dt0, dtype, force_coop=False, ts=None):
    ts = [0.0, 0.11 * t1, 0.5 * t1, 0.52 * t1, t1] if ts is None else ts
    ctl = P.PIDController(rtol=1e-4, atol=1e-6) if dtype is np.float32 else P.PIDController(rtol=1e-6, atol=1e-9, pcoeff=0.3, icoeff=0.4)
    eng = P.HipEngine()
    if force_coop:
        eng.set_small_persist(2)
    got = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0.astype(dtype), saveat=P.SaveAt(t0=True, ts=ts, t1=True), stepsize_controller=ctl, engine=eng)
    eng.close()
    want = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0.astype(np.float64), saveat=P.SaveAt(t0=True, ts=ts, t1=True),
                         stepsize_controller=ctl, engine=OracleEngine())
    return got, want


def _check(got, want, y0, dtype, min_steps=5, f32_abs=2e-5, f32_rel=2e-3):
    assert got.stats["kernel"].startswith("tsit5_coop"), got.stats["kernel"]
    np.testing.assert_array_equal(got.ts, want.ts)
    assert got.ys.shape == want.ys.shape and got.ys.dtype == dtype
    assert np.all(np.isfinite(got.ys))
    inc_g, inc_w = got.ys.astype(np.float64) - y0, want.ys - y0
    if dtype is np.float64:
        assert got.stats["num_accepted_steps"] == want.stats["num_accepted_steps"], (got.stats, want.stats)
        assert got.stats["num_rejected_steps"] == want.stats["num_rejected_steps"], (got.stats, want.stats)
        assert rel_l2(inc_g[2:], inc_w[2:]) < 1e-9, rel_l2(inc_g[2:], inc_w[2:])
    else:
        assert abs(got.stats["num_accepted_steps"] - want.stats["num_accepted_steps"]) <= max(3, want.stats["num_accepted_steps"] // 10)
        # the same step sequence: fp32 rounding only.  A borderline decision that fell the other way on the rounding noise
        # of the fp32 error estimate puts the two runs on different step sequences: both then sit within the
        # controller's tolerance (rtol 1e-4 per step) of the true solution, not within rounding of each other
        same = (got.stats["num_accepted_steps"], got.stats["num_rejected_steps"]) == (want.stats["num_accepted_steps"], want.stats["num_rejected_steps"])
        assert np.max(np.abs(got.ys - want.ys)) < (f32_abs if same else 2e-3), (same, float(np.max(np.abs(got.ys - want.ys))))
        assert rel_l2(inc_g[2:], inc_w[2:]) < (f32_rel if same else 2e-2), rel_l2(inc_g[2:], inc_w[2:])
    assert got.stats["num_accepted_steps"] > min_steps


# periodic Cahn-Hilliard / Allen-Cahn: a grid the single-workgroup kernel also takes (forced onto several workgroups),
# grids beyond it, ragged tile splits (100 = 4 x 25, 72 x 120), a single tile row (24 x 160)
@pytest.mark.parametrize("shape,force", [((64, 64), True), ((100, 100), False), ((128, 128), False), ((72, 120), False),
                                         ((24, 160), True)], ids=lambda v: str(v))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["ch", "ac"])
def test_periodic_multi_workgroup_solve_vs_oracle_driven_loop(shape, force, dtype, kind):
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    rng = np.random.default_rng(1000 * nx + ny)
    if kind == "ch":
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny)), 0.05, 0.95)
        t1, dt0 = 2e-5, 1e-7
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
        y0 = 0.3 * rng.standard_normal((nx, ny))
        t1, dt0 = 0.05, 1e-4
    y0 = y0.astype(dtype).astype(np.float64)
    got, want = _solve_pair(eq, y0, t1, dt0, dtype, force_coop=force)
    _check(got, want, y0, dtype)


THETA_QUADRATIC = lambda t: 34.9065850398866 * t**2 - 10.4719755119660 * t + np.pi / 2  # noqa: E731  (notebooks/smooth_boundary.ipynb:262)


@pytest.mark.parametrize("theta,flux", [(lambda t: np.pi / 2.0, lambda t: 0.0), (THETA_QUADRATIC, lambda t: 0.02 * (1.0 + 3.0 * t))],
                         ids=["notebook-constants", "theta(t)-flux(t)"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind,shape", [("ch", (100, 100)), ("ac", (64, 128)), ("ch", (60, 44))])
def test_smoothed_boundary_solve_vs_oracle_driven_loop(kind, shape, dtype, theta, flux):
    """the featured notebook's equation and grid (cahn_hilliard.py:204-289, allen_cahn.py:88-159): theta(t) / flux(t) are
    evaluated INSIDE the kernel at the stage times the controller chooses (constants and polynomials in t)"""
    psi = sbm_psi(*shape)
    dom = sbm_domain(P, psi)
    rng = np.random.default_rng(7)
    y0 = np.clip(0.5 + 0.1 * rng.standard_normal(shape), 0.1, 0.9).astype(dtype).astype(np.float64)
    if kind == "ac":
        eq = P.AllenCahn2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], theta)
        t1, dt0 = 1.0, 1e-3
    else:
        eq = P.CahnHilliard2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], theta, flux)
        t1, dt0 = 0.3, 1e-4
    got, want = _solve_pair(eq, y0, t1, dt0, dtype)
    _check(got, want, y0, dtype, min_steps=3)


def test_smoothed_boundary_non_polynomial_theta_stays_host_driven():
    """a theta(t) the kernel cannot evaluate itself (not a polynomial in t) keeps the host-driven trial / commit loop"""
    psi = sbm_psi(64, 64)
    dom = sbm_domain(P, psi)
    rng = np.random.default_rng(3)
    y0 = np.clip(0.5 + 0.1 * rng.standard_normal((64, 64)), 0.1, 0.9)
    eq = P.CahnHilliard2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], lambda t: 1.2 + 0.3 * np.sin(40.0 * t), lambda t: 0.0)
    sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, 2e-3, 1e-4, y0, stepsize_controller=P.PIDController(rtol=1e-6, atol=1e-9))
    assert not sol.stats["kernel"].startswith("tsit5_coop"), sol.stats["kernel"]
    want = P.diffeqsolve(eq, P.Tsit5(), 0.0, 2e-3, 1e-4, y0, stepsize_controller=P.PIDController(rtol=1e-6, atol=1e-9), engine=OracleEngine())
    assert sol.stats["num_accepted_steps"] == want.stats["num_accepted_steps"]
    assert rel_l2(sol.ys[-1] - y0, want.ys[-1] - y0) < 1e-9


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(64, 64), (128, 128), (40, 88)])
def test_advection_diffusion_solve_vs_oracle_driven_loop(shape, dtype):
    """notebooks/run_advection_diffusion.ipynb:31-84: Gaussian-potential-gradient velocity, Tsit5 + PID (steady
    velocity: the face fields sit in LDS for the whole solve)"""
    nx, ny = shape
    dom = P.Domain((nx, ny), ((0.0, 0.02 * nx), (0.0, 0.02 * ny)), "dimensionless")

    def velocity(t, x, y):
        g = np.exp(-((x - 0.4) ** 2 + (y - 0.4) ** 2) / (2 * 0.01))
        return -0.1 * (x - 0.4) / 0.01 * g, -0.1 * (y - 0.4) / 0.01 * g

    eq = P.AdvectionDiffusion2D(dom, velocity, 0.1, time_dependent=False)
    rng = np.random.default_rng(0)
    y0 = (0.5 + 0.01 * rng.standard_normal(shape)).astype(dtype).astype(np.float64)
    got, want = _solve_pair(eq, y0, 2e-3, 1e-5, dtype)
    # fp32: the white-noise start (the notebook's) keeps the controller at the explicit stability limit, where the damping
    # of the highest modes depends steeply on the step size: the rounding noise of the fp32 error estimate moves dt by
    # ~1e-3 relative, which shows as ~1e-4 (1 % of the 0.01 noise amplitude) at the EARLY save points and is gone at
    # t1, where those modes have decayed -- both runs are within the controller's rtol of the true solution throughout
    _check(got, want, y0, dtype, f32_abs=5e-4, f32_rel=2e-2, min_steps=3)
    assert np.max(np.abs(got.ys[-1] - want.ys[-1])) < (1e-10 if dtype is np.float64 else 2e-5)
    # conservative flux form: the mean does not move (run_advection_diffusion.ipynb:85-86)
    assert abs(got.ys[-1].astype(np.float64).mean() - y0.mean()) < (1e-12 if dtype is np.float64 else 1e-6)


@pytest.mark.parametrize("batch", [3, 19])
def test_every_environment_runs_its_own_controller(batch):
    """several environments in one call (PIDController(per_environment=True)): each gets its own workgroups, barrier and
    controller; more environments than one launch holds (16 workgroups each, 16 environments per launch) go in waves.
    Bitwise equal to solving them one by one."""
    n = 100
    psi = sbm_psi(n, n)
    dom = sbm_domain(P, psi)
    eq = P.CahnHilliard2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], THETA_QUADRATIC, lambda t: 0.01)
    rng = np.random.default_rng(5)
    amp = np.linspace(0.02, 0.12, batch)
    y0 = np.stack([np.clip(0.5 + a * rng.standard_normal((n, n)), 0.1, 0.9) for a in amp]).astype(np.float32)
    ctl = P.PIDController(rtol=1e-4, atol=1e-6, per_environment=True)
    ts = [0.0, 0.004, 0.01]
    eng = P.HipEngine()
    sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.01, 1e-4, y0, saveat=P.SaveAt(ts=ts), stepsize_controller=ctl, engine=eng)
    assert sol.stats["kernel"].startswith("tsit5_coop"), sol.stats["kernel"]
    acc = sol.stats["num_accepted_steps"]
    assert len(acc) == batch
    for b in (0, batch // 2, batch - 1):
        one = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.01, 1e-4, y0[b], saveat=P.SaveAt(ts=ts), stepsize_controller=P.PIDController(rtol=1e-4, atol=1e-6),
                            engine=eng)
        np.testing.assert_array_equal(one.ys, sol.ys[:, b])
        assert one.stats["num_accepted_steps"] == acc[b]
    eng.close()


def test_max_steps_and_argument_checks():
    dom = std_domain(P, 100, 100)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    rng = np.random.default_rng(1)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((100, 100)), 0.05, 0.95)
    ctl = P.PIDController(rtol=1e-6, atol=1e-9)
    sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, 1e-3, 1e-7, y0, stepsize_controller=ctl, max_steps=7, throw=False)
    assert sol.stats["kernel"].startswith("tsit5_coop") and sol.stats["num_steps"] == 7
    with pytest.raises(RuntimeError, match="max_steps"):
        P.diffeqsolve(eq, P.Tsit5(), 0.0, 1e-3, 1e-7, y0, stepsize_controller=ctl, max_steps=7, throw=True)
