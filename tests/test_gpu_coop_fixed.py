"""Fixed-step explicit Euler / RK4 for ONE environment on several compute units (csrc/stencil_coop_adaptive.hpp, MODE 1;
VERDICT r3 Weak #6): the communication-avoiding tiling of the adaptive kernel -- tile + 8 halo in LDS, as many substeps per
exchange as the halo pays for, one neighbour exchange per round -- for single environments of 96^2 - 320^2 cells, which the
tiled kernels advance at two dependent launches per substep whatever their size.

Gates: the CPU oracle on the same inputs (fp64 1e-9 of the increment; fp32 the state's rounding), and the tiled path on the
same GPU (fp64: rounding of a different association only)."""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import MOB, MU, SBM_F, SBM_FLUX, SBM_THETA, inc_tol_f32, rel_l2, sbm_domain, sbm_psi, std_domain

pytestmark = pytest.mark.gpu


def _advance(eq, y0, integ, dt, n, opt, t0=0.0, kappas=None):
    eng = P.HipEngine()
    eng.set_small_persist(opt)
    eng.configure(dtype=y0.dtype, batch=y0.shape[0], **eq._engine_problem())
    eq._engine_upload(eng, t0, dt * n)
    if kappas is not None:
        eng.set_env_params(0, kappa=kappas)
    eng.set_state(y0)
    eng.advance(integ, dt, n, t0)
    out, kern = eng.get_state(), eng.last_kernel
    eng.close()
    return out, kern


def _check(got, ref, y0, dtype, kern):
    inc_g, inc_w = got.astype(np.float64) - y0, ref - y0
    if dtype is np.float64:
        assert rel_l2(inc_g, inc_w) < 1e-9, (kern, rel_l2(inc_g, inc_w))
    else:
        assert np.max(np.abs(got - ref)) < 2e-6, (kern, float(np.max(np.abs(got - ref))))
        assert rel_l2(inc_g, inc_w) < inc_tol_f32(ref, y0), (kern, rel_l2(inc_g, inc_w))


# square, a ragged split (100 = 7 x 14.3), non-square, a grid on several XCDs, one the one-CU kernel also takes (forced)
@pytest.mark.parametrize("shape", [(128, 128), (100, 100), (72, 120), (256, 256), (64, 64)], ids=str)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["ch", "ac"])
def test_periodic_fixed_step_vs_oracle_and_tiled_path(shape, dtype, kind):
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    rng = np.random.default_rng(nx * 1000 + ny)
    if kind == "ch":
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        y0 = np.clip(0.5 + 0.05 * rng.standard_normal((2, nx, ny)), 0.05, 0.95).astype(dtype)
        f = lambda kap: (lambda t, u: O.ch_rhs_fd(u, *dom.dx, kap, MU["regsol"], MOB["c1mc"]))
        dt = 2e-7
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
        y0 = (0.1 * rng.standard_normal((2, nx, ny))).astype(dtype)
        f = lambda kap: (lambda t, u: O.ac_rhs_fd(u, *dom.dx, kap, MU["cubic"], MOB["one"]))
        dt = 5e-5
    kappas = [0.002, 0.0026]  # per-environment control values ride along
    # RK4: 9 substeps = 9 rounds (CH) / 4.5 (AC: two substeps per exchange, the last round half full); Euler: 11 = 2.75 / 1.4 rounds
    for integ, step, n, tag in ((L.INT_RK4, O.rk4_step, 9, "rk4_coop"), (L.INT_EULER, O.euler_step, 11, "euler_coop")):
        got, kern = _advance(eq, y0, integ, dt, n, 2, kappas=kappas)
        assert kern.startswith(tag) and "workgroups" in kern, kern
        tiled, kern_t = _advance(eq, y0, integ, dt, n, -1, kappas=kappas)
        assert "coop" not in kern_t, kern_t
        for b in range(2):
            ref = y0[b].astype(np.float64)
            for i in range(n):
                ref = step(f(kappas[b]), i * dt, ref, dt)
            _check(got[b], ref, y0[b].astype(np.float64), dtype, kern)
        if dtype is np.float64:
            assert rel_l2(got - y0, tiled - y0) < 1e-11, (kern, kern_t)
        else:
            assert np.max(np.abs(got - tiled)) < 5e-7, (kern, kern_t, float(np.max(np.abs(got - tiled))))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind,shape", [("ch", (100, 100)), ("ac", (72, 120))])
@pytest.mark.parametrize("integ", ["rk4", "euler"])
def test_smoothed_boundary_fixed_step_with_time_dependent_contact_angle(kind, shape, dtype, integ):
    """theta(t), flux(t) (a quadratic / linear in t: notebooks/smooth_boundary.ipynb:262) evaluated by the kernel at its own
    stage times t, t + dt / 2, t + dt of every substep of a round"""
    psi = sbm_psi(*shape)
    dom = sbm_domain(P, psi)
    rng = np.random.default_rng(5)
    y0 = np.clip(0.5 + 0.1 * rng.standard_normal((1,) + shape), 0.1, 0.9).astype(dtype)
    if kind == "ac":
        eq = P.AllenCahn2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA)
        f = lambda t, u: O.ac_sbm_rhs(u, psi, 1.0, 1.0, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA(t), eq.left_half)
        dt = 2e-2
    else:
        eq = P.CahnHilliard2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA, SBM_FLUX)
        f = lambda t, u: O.ch_sbm_rhs(u, psi, 1.0, 1.0, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA(t), SBM_FLUX(t), eq.left_half)
        dt = 2e-3
    t0, n = 0.03, 9
    code, step = (L.INT_RK4, O.rk4_step) if integ == "rk4" else (L.INT_EULER, O.euler_step)
    got, kern = _advance(eq, y0, code, dt, n, 2, t0=t0)
    assert kern.startswith(integ + "_coop") and "SBM" in kern, kern
    ref = y0[0].astype(np.float64)
    for i in range(n):
        ref = step(f, t0 + i * dt, ref, dt)
    if dtype is np.float64:
        assert rel_l2(got[0] - y0[0], ref - y0[0]) < 1e-9
    else:
        assert np.max(np.abs(got[0] - ref)) < 5e-6 and rel_l2(got[0].astype(np.float64) - y0[0], ref - y0[0]) < 2e-3


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_advection_diffusion_fixed_step(dtype):
    nx, ny = 96, 128
    dom = P.Domain((nx, ny), ((0.0, 0.02 * nx), (0.0, 0.02 * ny)), "dimensionless")

    def velocity(t, x, y):
        g = np.exp(-((x - 0.9) ** 2 + (y - 1.2) ** 2) / (2 * 0.05))
        return -0.1 * (x - 0.9) / 0.05 * g, -0.1 * (y - 1.2) / 0.05 * g

    eq = P.AdvectionDiffusion2D(dom, velocity, 0.1, time_dependent=False)
    rng = np.random.default_rng(2)
    y0 = (0.5 + 0.01 * rng.standard_normal((1, nx, ny))).astype(dtype)
    for code, n, tag in ((L.INT_RK4, 10, "rk4_coop"), (L.INT_EULER, 17, "euler_coop")):
        got, kern = _advance(eq, y0, code, 2e-4, n, 2)
        assert kern.startswith(tag) and ",AD," in kern, kern
        want, kern_t = _advance(eq, y0, code, 2e-4, n, -1)
        assert "coop" not in kern_t
        if dtype is np.float64:
            assert rel_l2(got - y0, want - y0) < 1e-11
        else:
            assert np.max(np.abs(got - want)) < 5e-7
        assert abs(float(got.astype(np.float64).mean() - y0.astype(np.float64).mean())) < (1e-12 if dtype is np.float64 else 1e-6)


@pytest.mark.parametrize("dtype,shape", [(np.float32, (96, 96)), (np.float64, (64, 64))], ids=["f32-96", "f64-64"])
def test_sixteen_environments_in_one_launch(dtype, shape):
    """the automatic choice for a batch of 16: 4 x 4 workgroups per environment, two environments per XCD, all in ONE launch
    (block -> (environment, tile) map over rows of XCD slots); per-environment kappa; first, middle and last environment
    against the oracle"""
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    rng = np.random.default_rng(16)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((16, nx, ny)), 0.05, 0.95).astype(dtype)
    kappas = [0.002 + 1e-5 * b for b in range(16)]
    got, kern = _advance(eq, y0, L.INT_RK4, 2e-7, 9, 0, kappas=kappas)
    assert kern.startswith("rk4_coop") and "4x4 workgroups" in kern, kern
    for b in (0, 7, 8, 15):
        ref = y0[b].astype(np.float64)
        for i in range(9):
            ref = O.rk4_step(lambda t, u: O.ch_rhs_fd(u, *dom.dx, kappas[b], MU["regsol"], MOB["c1mc"]), 0.0, ref, 2e-7)
        _check(got[b], ref, y0[b].astype(np.float64), dtype, kern)


def test_tiles_narrower_than_the_halo_wait_for_every_workgroup(monkeypatch):
    """tiles of 6 x 6 cells under an 8-cell halo: a ring reaches two tiles away, so the exchange waits for every workgroup
    of the environment instead of the 8 neighbours (PDEOPT_COOP_TILE forces the tile edge)"""
    monkeypatch.setenv("PDEOPT_COOP_TILE", "6")
    nx, ny = 48, 60
    dom = std_domain(P, nx, ny)
    rng = np.random.default_rng(9)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((1, nx, ny)), 0.05, 0.95)
    got, kern = _advance(eq, y0, L.INT_RK4, 2e-7, 9, 2)
    assert kern.startswith("rk4_coop") and "8x10 workgroups" in kern, kern
    ref = y0[0]
    for i in range(9):
        ref = O.rk4_step(lambda t, u: O.ch_rhs_fd(u, *dom.dx, 0.002, MU["regsol"], MOB["c1mc"]), 0.0, ref, 2e-7)
    assert rel_l2(got[0] - y0[0], ref - y0[0]) < 1e-9


def test_auto_policy():
    """one mid-sized environment and enough substeps: several CUs per environment; a batch the tiled kernels sweep in one
    launch, a few substeps, a grid of the one-CU kernel's range or a caller-chosen tiled knob: the other paths"""
    dom = std_domain(P, 128, 128)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    rng = np.random.default_rng(0)

    def kernel(batch, n, shape=(128, 128), knob=None):
        d = std_domain(P, *shape)
        e = P.CahnHilliard2DPeriodic(d, 0.002, MU["regsol"], MOB["c1mc"])
        y = np.clip(0.5 + 0.05 * rng.standard_normal((batch,) + shape), 0.05, 0.95).astype(np.float32)
        eng = P.HipEngine()
        if knob:
            knob(eng)
        eng.configure(dtype=y.dtype, batch=batch, **e._engine_problem())
        eng.set_state(y)
        eng.advance(L.INT_RK4, 2e-7, n)
        k = eng.last_kernel
        assert np.all(np.isfinite(eng.get_state()))
        eng.close()
        return k

    assert kernel(1, 100).startswith("rk4_coop"), kernel(1, 100)
    assert kernel(2, 100).startswith("rk4_coop")
    assert not kernel(64, 100).startswith("rk4_coop")
    assert not kernel(1, 4).startswith("rk4_coop")
    assert kernel(1, 100, (64, 64)).startswith("small_persist")
    assert not kernel(1, 100, (256, 256)).startswith("rk4_coop")  # the tiled whole-substep kernel is as fast from ~200^2 on
    assert not kernel(1, 100, knob=lambda e: e.set_fuse_stages(1)).startswith("rk4_coop")
