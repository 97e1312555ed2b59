"""The whole-environment-step kernel for LDS-resident grids (csrc/stencil_small.hpp): all n substeps
of Euler / RK4 in ONE launch, one workgroup per environment -- the regime of the reference's own tests and notebooks
(32^2 ... 128^2: tests/test_solvers.py:25,68,145; the loop at pde_opt/pde_env.py:293-303).

Checked against (i) the CPU oracle on the same inputs (the parity gate proper) and (ii) the tiled stage-pair path on
the same GPU, which it restates expression for expression: <= 1 ulp of the state (bitwise on most cells)."""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import MOB, MU, inc_tol_f32, rel_l2, std_domain

pytestmark = pytest.mark.gpu

# (nx, ny): one vector per thread; two; the 512-thread form (3-8 vectors per thread); non-square; not a power of two;
# a single row of vectors; a "1-D" run
SHAPES = [(32, 32), (64, 64), (64, 128), (128, 128), (100, 100), (96, 40), (24, 36), (8, 8), (256, 4), (1, 64)]


def _problem(kind, nx, ny, dtype, closures):
    dom = std_domain(P, nx, ny)
    rng = np.random.default_rng(nx * 1000 + ny)
    if kind == "ch":
        mu, mob = closures
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU[mu], MOB[mob])
        y0 = np.clip(0.5 + 0.05 * rng.standard_normal((3, nx, ny)), 0.05, 0.95).astype(dtype)
        f = lambda b, kap: (lambda t, u: O.ch_rhs_fd(u, *dom.dx, kap, MU[mu], MOB[mob]))
        dt = 2e-7 if mob == "c1mc" else 2e-8  # D = 1 + c^2 > 1: the explicit stability limit of the biharmonic is 5x tighter
    else:
        mu, mob = closures
        eq = P.AllenCahn2DPeriodic(dom, 0.002, MU[mu], MOB[mob])
        y0 = (0.5 + 0.1 * rng.standard_normal((3, nx, ny))).astype(dtype) if mu == "regsol" else (0.1 * rng.standard_normal((3, nx, ny))).astype(dtype)
        if mu == "regsol":
            y0 = np.clip(y0, 0.05, 0.95)
        f = lambda b, kap: (lambda t, u: O.ac_rhs_fd(u, *dom.dx, kap, MU[mu], MOB[mob]))
        dt = 5e-5
    return eq, y0, f, dt


def _run(eq, y0, integ, dt, n, small, kappas=None):
    eng = P.HipEngine()
    eng.set_small_persist(1 if small else -1)
    if not small:
        eng.set_fuse_stages(1)  # stage pairs (Allen-Cahn: not the single-pass kernel, which folds its arithmetic differently)
    eng.configure(dtype=y0.dtype, batch=y0.shape[0], **eq._engine_problem())
    if kappas is not None:
        eng.set_env_params(0, kappa=kappas)
    eng.set_state(y0)
    eng.advance(integ, dt, n)
    out, kern = eng.get_state(), eng.last_kernel
    eng.close()
    return out, kern


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{a}x{b}" for a, b in SHAPES])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind,closures", [("ch", ("regsol", "c1mc")), ("ch", ("cubic", "one_plus_sq")), ("ac", ("cubic", "one"))])
def test_whole_step_kernel_vs_oracle_and_tiled_path(shape, dtype, kind, closures):
    nx, ny = shape
    if dtype is np.float64 and nx * ny > 8192:
        pytest.skip("fp64: stage input + chemical potential of > 8192 cells do not fit 8 vectors per thread")
    if dtype is np.float64 and ny % 2 or dtype is np.float32 and ny % 4:
        pytest.skip("ny must be a multiple of the 16-byte vector")
    eq, y0, f, dt = _problem(kind, nx, ny, dtype, closures)
    kappas = [0.002, 0.0025, 0.0015]  # per-environment control values ride along
    for integ, step, n in ((L.INT_RK4, O.rk4_step, 7), (L.INT_EULER, O.euler_step, 5)):
        got, kern = _run(eq, y0, integ, dt, n, True, kappas)
        assert kern.startswith("small_persist"), kern
        for b in range(3):
            ref = y0[b].astype(np.float64)
            for i in range(n):
                ref = step(f(b, kappas[b]), i * dt, ref, dt)
            inc_g, inc_w = got[b].astype(np.float64) - y0[b], ref - y0[b]
            if dtype is np.float64:
                assert rel_l2(inc_g, inc_w) < 1e-9, (kern, b, rel_l2(inc_g, inc_w))
            else:
                assert np.max(np.abs(got[b] - ref)) < 1e-6, (kern, b)
                assert rel_l2(inc_g, inc_w) < inc_tol_f32(ref, y0[b]), (kern, b, rel_l2(inc_g, inc_w))
        # the tiled stage-pair path (where it covers the shape) evaluates the same expressions.  Cahn-Hilliard: every
        # ambiguous contraction of the flux arithmetic is pinned (stencil_fused.hpp), the two agree to an ulp of the
        # state and on > 99 % of the cells bitwise.  Allen-Cahn: lap = dxx / hx^2 + dyy / hy^2 is a sum of two products
        # the compiler may contract either way round per kernel, and kappa lap cancels against mu_h: rounding-level
        # differences of the INCREMENT.
        tiled, kern_t = _run(eq, y0, integ, dt, n, False, kappas)
        if "pair" in kern_t and kind == "ch":
            ulp = np.spacing(np.abs(tiled).astype(dtype))
            assert np.max(np.abs(got - tiled) / ulp) <= 1.0, (kern, kern_t, float(np.max(np.abs(got - tiled) / ulp)))
            assert np.mean(got != tiled) < 0.01, (kern, kern_t, float(np.mean(got != tiled)))
        elif "pair" in kern_t:
            d = rel_l2(got.astype(np.float64) - y0, tiled.astype(np.float64) - y0)
            assert d < (1e-12 if dtype is np.float64 else 1e-5), (kern, kern_t, d)


def test_auto_policy_and_api_path():
    """auto: grids up to 4096 cells take the whole-step kernel (PDEEnv / diffeqsolve get it without asking), a single
    128^2 environment stays on the tiled kernels, 192 of them do not; -1 switches it off; shapes it cannot hold fall
    through silently to the other paths"""
    def kernel_for(nx, ny, batch, opt=0, integ=None, n=6):
        dom = std_domain(P, nx, ny)
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        eng = P.HipEngine()
        eng.set_small_persist(opt)
        y0 = np.full((batch, nx, ny), 0.5, np.float32)
        sol = P.diffeqsolve(eq, integ or P.RK4(), 0.0, n * 2e-7, 2e-7, y0, engine=eng)
        k = sol.stats["kernel"]
        eng.close()
        return k

    assert kernel_for(64, 64, 1).startswith("small_persist")
    assert kernel_for(32, 32, 5).startswith("small_persist")
    assert "small_persist" not in kernel_for(128, 128, 1)
    assert "small_persist" not in kernel_for(128, 128, 16)
    assert kernel_for(128, 128, 192).startswith("small_persist")
    assert kernel_for(128, 128, 1, opt=1).startswith("small_persist")
    assert "small_persist" not in kernel_for(64, 64, 1, opt=-1)
    assert "small_persist" not in kernel_for(256, 256, 1, opt=1)       # 512 KB: not LDS-resident
    assert "small_persist" not in kernel_for(64, 64, 1, integ=P.Tsit5())  # Euler / RK4 only
    assert "small_persist" not in kernel_for(64, 64, 1, n=1)            # a single substep: nothing to keep resident


def test_whole_step_kernel_is_deterministic_and_batch_independent():
    dom = std_domain(P, 64, 64)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    rng = np.random.default_rng(3)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((300, 64, 64)), 0.05, 0.95).astype(np.float32)  # > 256 workgroups
    a, _ = _run(eq, y0, L.INT_RK4, 2e-7, 20, True)
    b, _ = _run(eq, y0, L.INT_RK4, 2e-7, 20, True)
    np.testing.assert_array_equal(a, b)
    solo, _ = _run(eq, y0[17:18], L.INT_RK4, 2e-7, 20, True)
    np.testing.assert_array_equal(solo[0], a[17])
    two, _ = _run(eq, y0, L.INT_RK4, 2e-7, 10, True)
    eng = P.HipEngine()
    eng.set_small_persist(1)
    eng.configure(dtype=np.float32, batch=300, **eq._engine_problem())
    eng.set_state(y0)
    eng.advance(L.INT_RK4, 2e-7, 10)
    eng.advance(L.INT_RK4, 2e-7, 10)  # two calls of 10 == one of 20: the state round-trips through global memory exactly
    np.testing.assert_array_equal(eng.get_state(), a)
    eng.close()
