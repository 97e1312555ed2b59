"""The in-kernel adaptive solve (csrc/stencil_small_adaptive.hpp, pdeopt_tsit5_solve_small): Tsit5 trial steps, the
scaled RMS error norm, diffrax.PIDController's update, accept / reject and the dense output for SaveAt(ts) inside ONE
launch, one workgroup per environment -- the loop the reference runs through diffrax.diffeqsolve
(pde_opt/pde_model.py:100-118; tests/test_solvers.py:64-104 is the 256 x 1 Allen-Cahn case below).

The gate: the same solve driven step by step on the CPU oracle (tests/fake_engine.py: OracleEngine under the
package's own host loop) -- states at the save points, and the controller's accept / reject sequence."""
import numpy as np
import pytest

import pde_opt_amd as P
from fake_engine import OracleEngine
from util import MOB, MU, rel_l2, std_domain

pytestmark = pytest.mark.gpu

# one vector per thread (<= 512 vectors), two, four; the reference's 1-D column; a row; not a power of two; vector counts
# that are no multiple of the 512 threads with two / four vectors per thread (threads past the end redo the last vector
# from another wave than its owner: the FSAL read of k7 needs its barrier)
SHAPES = [(32, 32), (64, 64), (64, 128), (256, 1), (1, 64), (24, 36), (48, 48), (64, 96)]


def _case(kind, nx, ny, dtype, seed=0):
    dom = std_domain(P, nx, ny)
    rng = np.random.default_rng(1000 * nx + ny + seed)
    if kind == "ch":
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(dtype)
        t1, dt0 = 2e-5, 1e-7
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
        y0 = (0.3 * rng.standard_normal((nx, ny))).astype(dtype)
        t1, dt0 = 0.05, 1e-4
    return eq, y0, t1, dt0


def _fits(nx, ny, dtype):
    v = 4 if dtype is np.float32 else 2
    row = nx if ny == 1 else ny
    return row % v == 0 and nx * ny // v <= 2048


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{a}x{b}" for a, b in SHAPES])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["ch", "ac"])
def test_in_kernel_solve_vs_oracle_driven_loop(shape, dtype, kind):
    nx, ny = shape
    if not _fits(nx, ny, dtype):
        pytest.skip("more than 2048 vectors per environment")
    eq, y0, t1, dt0 = _case(kind, nx, ny, dtype)
    ts = [0.0, 0.11 * t1, 0.5 * t1, 0.52 * t1, t1]
    ctl = P.PIDController(rtol=1e-4, atol=1e-6) if dtype is np.float32 else P.PIDController(rtol=1e-6, atol=1e-9, pcoeff=0.3, icoeff=0.4)
    eng = P.HipEngine()
    eng.set_small_persist(1)  # THIS kernel (auto hands grids of three or more vectors per thread to the multi-workgroup one)
    got = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0, saveat=P.SaveAt(t0=True, ts=ts, t1=True), stepsize_controller=ctl, engine=eng)
    eng.close()
    assert got.stats["kernel"].startswith("small_tsit5"), got.stats["kernel"]
    want = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0.astype(np.float64), saveat=P.SaveAt(t0=True, ts=ts, t1=True),
                         stepsize_controller=ctl, engine=OracleEngine())
    np.testing.assert_array_equal(got.ts, want.ts)
    assert got.ys.shape == want.ys.shape and got.ys.dtype == dtype
    assert np.all(np.isfinite(got.ys))
    inc_g, inc_w = got.ys.astype(np.float64) - y0, want.ys - y0
    if dtype is np.float64:
        # the same arithmetic in another summation order: the controller takes the same decisions
        assert got.stats["num_accepted_steps"] == want.stats["num_accepted_steps"]
        assert got.stats["num_rejected_steps"] == want.stats["num_rejected_steps"]
        assert rel_l2(inc_g[2:], inc_w[2:]) < 1e-9, rel_l2(inc_g[2:], inc_w[2:])
    else:
        # fp32 error estimates carry rounding noise: a borderline decision may fall the other way, after which the two
        # runs sit on different step sequences -- both within the controller's tolerance of the true solution
        assert abs(got.stats["num_accepted_steps"] - want.stats["num_accepted_steps"]) <= max(3, want.stats["num_accepted_steps"] // 10)
        assert np.max(np.abs(got.ys - want.ys)) < 2e-5, float(np.max(np.abs(got.ys - want.ys)))
        assert rel_l2(inc_g[2:], inc_w[2:]) < 2e-3, rel_l2(inc_g[2:], inc_w[2:])
    assert got.stats["num_accepted_steps"] > 5


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind,shape", [("ch", (64, 64)), ("ac", (256, 1)), ("ch", (32, 64))])
def test_in_kernel_solve_vs_host_driven_loop_on_the_gpu(dtype, kind, shape):
    """the same solve through pdeopt_tsit5_trial / commit / dense under the host's controller (the path larger grids
    take): the same tableau, norm and controller in both"""
    eq, y0, t1, dt0 = _case(kind, *shape, dtype, seed=5)
    ts = np.linspace(0.0, t1, 9)
    ctl = P.PIDController(rtol=1e-4, atol=1e-6)
    out = []
    for opt in (1, -1):  # the single-workgroup kernel wherever it runs; the host-driven loop
        eng = P.HipEngine()
        eng.set_small_persist(opt)
        out.append(P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0, saveat=P.SaveAt(ts=ts), stepsize_controller=ctl, engine=eng))
        eng.close()
    a, b = out
    assert a.stats["kernel"].startswith("small_tsit5") and not b.stats["kernel"].startswith("small_tsit5"), (a.stats, b.stats)
    if dtype is np.float64:
        assert a.stats["num_accepted_steps"] == b.stats["num_accepted_steps"]
        assert a.stats["num_rejected_steps"] == b.stats["num_rejected_steps"]
        assert np.max(np.abs(a.ys - b.ys)) < 1e-11
    else:
        assert abs(a.stats["num_accepted_steps"] - b.stats["num_accepted_steps"]) <= max(3, b.stats["num_accepted_steps"] // 10)
        assert np.max(np.abs(a.ys - b.ys)) < 2e-5


def test_every_environment_runs_its_own_controller():
    """PIDController(per_environment=True) on a batch == the environments solved alone, bit for bit (one workgroup
    each, the same code); a shared step size (per_environment=False) stays on the host-driven loop"""
    nx, ny = 64, 64
    dom = std_domain(P, nx, ny)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    rng = np.random.default_rng(2)
    y0 = np.stack([0.05 * rng.standard_normal((nx, ny)), 0.9 * np.sign(rng.standard_normal((nx, ny))),
                   0.5 + 0.3 * rng.standard_normal((nx, ny))]).astype(np.float32)
    ts = [0.0, 0.013, 0.05, 0.2]
    ctl = dict(rtol=1e-4, atol=1e-6)
    eng = P.HipEngine()
    solo = [P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0[b], saveat=P.SaveAt(ts=ts, t1=True),
                          stepsize_controller=P.PIDController(**ctl), engine=eng) for b in range(3)]
    both = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(ts=ts, t1=True),
                         stepsize_controller=P.PIDController(**ctl, per_environment=True), engine=eng)
    assert both.stats["kernel"].startswith("small_tsit5")
    for b in range(3):
        np.testing.assert_array_equal(both.ys[:, b], solo[b].ys)
        assert both.stats["num_accepted_steps"][b] == solo[b].stats["num_accepted_steps"]
        assert both.stats["num_rejected_steps"][b] == solo[b].stats["num_rejected_steps"]
    assert len(set(both.stats["num_accepted_steps"])) > 1
    shared = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(ts=ts), stepsize_controller=P.PIDController(**ctl),
                           engine=eng)
    assert not shared.stats["kernel"].startswith("small_tsit5")
    assert np.max(np.abs(shared.ys - both.ys[:-1])) < 1e-4  # different step sequences, the same solution to tolerance
    eng.close()


def test_more_environments_than_compute_units():
    dom = std_domain(P, 32, 32)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    rng = np.random.default_rng(8)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((300, 32, 32)), 0.05, 0.95).astype(np.float32)
    eng = P.HipEngine()
    ctl = P.PIDController(rtol=1e-4, atol=1e-6, per_environment=True)
    a = P.diffeqsolve(eq, P.Tsit5(), 0.0, 1e-5, 1e-7, y0, stepsize_controller=ctl, engine=eng)
    b = P.diffeqsolve(eq, P.Tsit5(), 0.0, 1e-5, 1e-7, y0[280:283], stepsize_controller=ctl, engine=eng)
    eng.close()
    assert a.stats["kernel"].startswith("small_tsit5")
    np.testing.assert_array_equal(a.ys[:, 280:283], b.ys)
    assert np.all(np.isfinite(a.ys))


def test_step_budget():
    """max_steps counts trial steps; throw=True raises like the host loop, throw=False returns where the solve stopped:
    the save points reached, then (t, y) there"""
    eq, y0, t1, dt0 = _case("ac", 64, 64, np.float64)
    ts = list(np.linspace(0.0, t1, 6))
    ctl = P.PIDController(rtol=1e-6, atol=1e-9)
    eng = P.HipEngine()
    eng.set_small_persist(1)
    with pytest.raises(RuntimeError, match="max_steps=7"):
        P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0, saveat=P.SaveAt(ts=ts), stepsize_controller=ctl, max_steps=7, engine=eng)
    got = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0, saveat=P.SaveAt(ts=ts, t1=True), stepsize_controller=ctl, max_steps=7,
                        throw=False, engine=eng)
    eng.close()
    want = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0, saveat=P.SaveAt(ts=ts, t1=True), stepsize_controller=ctl, max_steps=7,
                         throw=False, engine=OracleEngine())
    assert got.stats["kernel"].startswith("small_tsit5")
    assert got.stats["num_steps"] == want.stats["num_steps"] == 7
    # the error estimate is a cancelling sum (sum_j e_j = 0): its relative rounding noise, ~1e-12 here, is the step
    # sizes' too
    np.testing.assert_allclose(got.ts, want.ts, rtol=1e-9)
    assert got.ts[-1] < t1
    assert np.max(np.abs(got.ys - want.ys)) < 1e-10


def test_controller_limits_and_pid_terms():
    """dtmax clips the step the controller asks for; a PID (not just I) controller takes the same decisions as the
    host's arithmetic"""
    eq, y0, t1, dt0 = _case("ac", 32, 32, np.float64)
    for ctl in (P.PIDController(rtol=1e-5, atol=1e-8, dtmax=2e-3), P.PIDController(rtol=1e-5, atol=1e-8, pcoeff=0.2, icoeff=0.5, dcoeff=0.1),
                P.PIDController(rtol=1e-5, atol=1e-8, dtmin=1e-3, factormax=3.0, safety=0.8)):
        eng = P.HipEngine()
        got = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0, stepsize_controller=ctl, engine=eng)
        eng.close()
        want = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0, stepsize_controller=ctl, engine=OracleEngine())
        assert got.stats["kernel"].startswith("small_tsit5")
        assert (got.stats["num_accepted_steps"], got.stats["num_rejected_steps"]) == (
            want.stats["num_accepted_steps"], want.stats["num_rejected_steps"]), (ctl, got.stats, want.stats)
        assert np.max(np.abs(got.ys - want.ys)) < 1e-11
    # dtmax = 2e-3 over t1 = 0.05: at least 25 accepted steps
    eng = P.HipEngine()
    got = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y0, stepsize_controller=P.PIDController(rtol=1e-2, atol=1e-2, dtmax=2e-3), engine=eng)
    eng.close()
    assert got.stats["num_accepted_steps"] >= 25


def test_c_abi_argument_checks_and_unsupported_problems():
    eq, y0, t1, dt0 = _case("ac", 32, 32, np.float32)
    eng = P.HipEngine()
    eng.configure(dtype=np.float32, batch=1, **eq._engine_problem())
    eng.set_state(y0[None])
    ctl = P.PIDController(rtol=1e-4, atol=1e-6)
    assert eng.tsit5_solve_small_supported()
    with pytest.raises(ValueError, match="ascending"):
        eng.tsit5_solve_small(0.0, t1, dt0, ctl, 100, [0.02, 0.01])
    with pytest.raises(ValueError, match="> t0"):
        eng.tsit5_solve_small(0.0, t1, dt0, ctl, 100, [0.0, 0.01])
    with pytest.raises(ValueError, match="max_steps"):
        eng.tsit5_solve_small(0.0, t1, dt0, ctl, 0)
    with pytest.raises(ValueError, match="dt0"):
        eng.tsit5_solve_small(0.0, t1, 0.0, ctl, 100)
    with pytest.raises(ValueError, match="rtol"):
        eng.tsit5_solve_small(0.0, t1, dt0, P.PIDController(rtol=0.0, atol=0.0), 100)
    np.testing.assert_array_equal(eng.get_state()[0], y0)  # refused calls leave the state alone
    saves, stats = eng.tsit5_solve_small(0.0, t1, dt0, ctl, 100000, [0.01, t1])
    assert stats[0]["status"] == 0 and stats[0]["t"] == t1 and stats[0]["saved"] == 2
    np.testing.assert_allclose(saves[1, 0], eng.get_state()[0], rtol=0, atol=1e-6)  # theta = 1: b_i(1) are the 5th-order weights
    eng.close()
    # 128^2 in fp32 is 4096 vectors: beyond one workgroup, taken by the multi-workgroup kernel since round 4
    # (stencil_coop_adaptive.hpp); Fourier derivatives have no in-kernel adaptive solve
    eng = P.HipEngine()
    dom = std_domain(P, 128, 128)
    big = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    eng.configure(dtype=np.float32, batch=1, **big._engine_problem())
    assert eng.tsit5_solve_small_supported()
    eng.set_small_persist(-1)  # ... unless the caller rules the whole-solve kernels out
    assert not eng.tsit5_solve_small_supported()
    eng.set_small_persist(0)
    fourier = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"], derivs="fourier")
    eng.configure(dtype=np.float32, batch=1, **fourier._engine_problem())
    assert not eng.tsit5_solve_small_supported()
    with pytest.raises(ValueError, match="in-kernel adaptive solve"):
        eng.tsit5_solve_small(0.0, t1, dt0, ctl, 100)
    eng.close()
