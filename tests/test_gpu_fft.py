"""The fused split-step (FFTs in LDS, csrc/strang_fused.hip + fft_lds.hpp) against the oracle and
against the rocFFT pipeline it replaces for power-of-two grids."""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import rel_l2

pytestmark = pytest.mark.gpu


def _setup(nx, ny, kinetic=True):
    dom = P.Domain((nx, ny), ((-12.0, 12.0), (-9.0, 9.0)), "dimensionless")
    eq = P.GPE2DTSControl(dom, 500.0, 0.2, lambda t, x, y: 0.05 * x - 0.02 * y, trap_factor=1.0, kinetic=kinetic)
    X, Y = dom.mesh()
    psi = np.exp(-(X**2 / 18.0 + Y**2 / 10.0)) * np.exp(0.3j * X - 0.2j * Y)
    psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * dom.dx[0] ** 2)
    return dom, eq, np.stack([psi.real, psi.imag], axis=-1), (X, Y)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(64, 64), (128, 64), (64, 256), (512, 128), (1024, 64), (128, 1024)])
@pytest.mark.parametrize("tscale", [1.0, -1j])
def test_fused_strang_vs_oracle(dtype, shape, tscale):
    nx, ny = shape
    dom, eq, y0, (X, Y) = _setup(nx, ny)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, tscale)
    n, dt = 4, 1e-3
    eng = P.HipEngine()
    sol = P.diffeqsolve(eq, solver, 0.0, n * dt, dt, y0.astype(dtype), engine=eng)
    assert eng.last_kernel == "strang_fused_lds_fft", eng.last_kernel
    eng.close()
    lights = 0.05 * X - 0.02 * Y
    b = lambda t, yy: O.gpe_b_terms(yy, X, Y, 500.0, 0.2, 1.0, lights)
    ref = y0
    for i in range(n):
        ref = O.strang_step(b, i * dt, ref, dt, eq.A_term, eq.dx, tscale)
    tol = 1e-11 if dtype is np.float64 else 2e-5
    assert rel_l2(sol.ys[-1], ref) < tol, rel_l2(sol.ys[-1], ref)


def test_fused_equals_rocfft_pipeline_and_batch():
    dom, eq, y0, _ = _setup(128, 128)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    yb = np.stack([y0, 0.5 * y0, y0[::-1].copy()])
    outs = {}
    for path in (L.PATH_AUTO, L.PATH_GENERIC):
        eng = P.HipEngine()
        eng.set_kernel_path(path)
        outs[path] = P.diffeqsolve(eq, solver, 0.0, 5e-3, 1e-3, yb, engine=eng).ys[-1]
        name = eng.last_kernel
        eng.close()
        assert ("lds_fft" in name) == (path == L.PATH_AUTO), name
    assert rel_l2(outs[L.PATH_AUTO], outs[L.PATH_GENERIC]) < 1e-12
    # every environment is renormalised separately
    dens = outs[L.PATH_AUTO][..., 0] ** 2 + outs[L.PATH_AUTO][..., 1] ** 2
    np.testing.assert_allclose(dens.sum(axis=(1, 2)) * dom.dx[0] ** 2, 1.0, rtol=1e-12)


def test_fused_zero_A_term_matches_committed_reference_behaviour():
    """the committed reference multiplies A_term by 0 (gross_pitaevskii.py:62): identity kinetic step"""
    dom, eq, y0, (X, Y) = _setup(64, 64, kinetic=False)
    assert np.abs(eq.A_term).max() == 0.0
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, -1j)
    sol = P.diffeqsolve(eq, solver, 0.0, 3e-3, 1e-3, y0)
    lights = 0.05 * X - 0.02 * Y
    b = lambda t, yy: O.gpe_b_terms(yy, X, Y, 500.0, 0.2, 1.0, lights)
    ref = y0
    for i in range(3):
        ref = O.strang_step(b, i * 1e-3, ref, 1e-3, eq.A_term, eq.dx, -1j)
    assert rel_l2(sol.ys[-1], ref) < 1e-12


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape,batch", [((64, 64), 2), ((128, 256), 3), ((1024, 128), 1), ((256, 1024), 2)])
def test_fused_imex_vs_oracle_and_rocfft(dtype, shape, batch):
    from util import MOB, MU, std_domain

    rng = np.random.default_rng(40)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((batch, nx, ny)), 0.05, 0.95).astype(dtype)
    n, dt = 6, 1e-6
    outs = {}
    for path in (L.PATH_AUTO, L.PATH_GENERIC):
        eng = P.HipEngine()
        if path != L.PATH_AUTO:  # "generic" = rocFFT's real<->hermitian plans
            eng.set_kernel_path(path)
        outs[path] = P.diffeqsolve(eq, solver, 0.0, n * dt, dt, y0, engine=eng).ys[-1]
        assert ("imex_fused_lds_fft" in eng.last_kernel) == (path == L.PATH_AUTO), eng.last_kernel
        eng.close()
    hx, hy = dom.dx
    sym = O.ch_fourier_symbol(nx, ny, hx, hy, 0.002)
    rhs = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, MU["regsol"], MOB["c1mc"])
    for b in range(batch):
        ref = y0[b].astype(np.float64)
        for i in range(n):
            ref = O.imex_step(rhs, i * dt, ref, dt, 0.5, sym)
        inc_ref = ref - y0[b].astype(np.float64)
        tol = 1e-9 if dtype is np.float64 else 5e-4
        assert rel_l2(outs[L.PATH_AUTO][b].astype(np.float64) - y0[b], inc_ref) < tol
        assert rel_l2(outs[L.PATH_GENERIC][b].astype(np.float64) - y0[b], inc_ref) < tol


def test_imex_full_size_properties():
    """BASELINE config 3 size (1024^2 fp32, odd batch so one complex field carries a single environment):
    the IMEX step conserves the mean (the k = 0 mode of the multiplier is 1 and the flux-form RHS sums
    to zero), pairs do not leak into each other (each environment equals its solo run), and the batch is
    linear in nothing it should not be (a constant field stays constant)."""
    from util import MOB, MU, std_domain

    rng = np.random.default_rng(7)
    n = 1024
    dom = std_domain(P, n, n)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((3, n, n)), 0.05, 0.95).astype(np.float32)
    y0[2] = 0.37  # constant field: rhs = 0, must stay put bit for bit
    eng = P.HipEngine()
    y1 = P.diffeqsolve(eq, solver, 0.0, 20e-6, 1e-6, y0, engine=eng).ys[-1]
    assert "imex_fused_lds_fft" in eng.last_kernel
    assert np.isfinite(y1).all()
    for b in range(2):
        assert abs(y1[b].astype(np.float64).mean() - y0[b].astype(np.float64).mean()) < 2e-7
        assert np.linalg.norm(y1[b] - y0[b]) > 0
    np.testing.assert_array_equal(y1[2], y0[2])
    solo = P.diffeqsolve(eq, solver, 0.0, 20e-6, 1e-6, y0[1], engine=eng).ys[-1]
    # env 1 rode in the imaginary part of pair 0 above and in the real part here
    assert rel_l2(y1[1].astype(np.float64) - y0[1], solo.astype(np.float64) - y0[1]) < 5e-4
    eng.close()


def test_strang_full_size_properties():
    """BASELINE config 4 size (512^2 c64): real-time Strang steps keep the norm at 1 (renormalised every
    step, with the reference's dx[0]^2 cell area, solvers.py:111) for every environment of the batch."""
    dom, eq, y0, _ = _setup(512, 512)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    yb = np.stack([y0, y0[::-1].copy()]).astype(np.float32)
    eng = P.HipEngine()
    fwd = P.diffeqsolve(eq, solver, 0.0, 10e-3, 1e-3, yb, engine=eng).ys[-1]
    assert eng.last_kernel == "strang_fused_lds_fft"
    dens = fwd[..., 0].astype(np.float64) ** 2 + fwd[..., 1].astype(np.float64) ** 2
    np.testing.assert_allclose(dens.sum(axis=(1, 2)) * dom.dx[0] ** 2, 1.0, rtol=2e-6)
    assert rel_l2(fwd, yb) > 1e-3  # it moved
    eng.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape,batch,group", [((64, 64), 5, 0), ((128, 256), 4, 3), ((1024, 64), 3, 2)])
def test_imex_per_environment_implicit_operator(dtype, shape, batch, group):
    """pdeopt_set_env_imex_scale: every environment integrates with its own kappa in the stencil AND in the implicit
    operator 1 / (1 + A dt kappa_b k^4) -- one environment per complex field, multiplier formed in the column pass"""
    from util import MOB, MU, std_domain

    rng = np.random.default_rng(41)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    kappas = 0.002 * (1.0 + 0.25 * np.arange(batch))
    eq = P.CahnHilliard2DPeriodic(dom, kappas[0], MU["regsol"], MOB["c1mc"])
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((batch, nx, ny)), 0.05, 0.95).astype(dtype)
    n, dt = 5, 1e-6
    eng = P.HipEngine()
    eng.set_group_envs(group)
    eng.configure(dtype=dtype, batch=batch, **eq._engine_problem())
    solver.configure_engine(eng, eq)
    eng.set_env_params(0, kappa=kappas)
    eng.set_env_imex_scale(0, kappas / kappas[0])
    eng.set_state(y0)
    eng.advance(L.INT_IMEX, dt, n)
    out = eng.get_state()
    assert "imex_fused_lds_fft" in eng.last_kernel
    hx, hy = dom.dx
    for b in range(batch):
        sym = O.ch_fourier_symbol(nx, ny, hx, hy, kappas[b])
        rhs = lambda t, u, kb=kappas[b]: O.ch_rhs_fd(u, hx, hy, kb, MU["regsol"], MOB["c1mc"])
        ref = y0[b].astype(np.float64)
        for i in range(n):
            ref = O.imex_step(rhs, i * dt, ref, dt, 0.5, sym)
        tol = 1e-9 if dtype is np.float64 else 5e-4
        assert rel_l2(out[b].astype(np.float64) - y0[b], ref - y0[b]) < tol, (b, rel_l2(out[b].astype(np.float64) - y0[b], ref - y0[b]))
    # all scales back to one: the paired transforms again, equal to a fresh engine
    eng.set_env_params(0, kappa=np.full(batch, kappas[0]))
    eng.set_env_imex_scale(0, np.ones(batch))
    eng.set_state(y0)
    eng.advance(L.INT_IMEX, dt, n)
    paired = eng.get_state()
    eng.close()
    eng2 = P.HipEngine()
    want = P.diffeqsolve(eq, solver, 0.0, n * dt, dt, y0, engine=eng2).ys[-1]
    eng2.close()
    np.testing.assert_array_equal(paired, want)
