"""Seeded random configurations of the hot path against the CPU oracle.

The targeted parity tests pin one feature at a time; the kernels, however, are selected from the COMBINATION of
equation, closure class, dtype, grid shape (tile-divisible, ragged, vector-misaligned, 1-D), batch, environment
groups, per-environment parameters, integrator, substep count (odd counts leave a trailing single stage pair /
Euler step) and launch options (hipGraph replay, per-stage / fused / generic kernels).  Each case below draws one
such combination from a fixed seed and compares the HIP result of every environment with oracle/np_oracle.py run
in fp64 on the same inputs.

Tolerances (stated as in tests/util.py): fp64 increments to 1e-9 relative (FMA / re-association, amplified by the
biharmonic's (1/h^2)^2 cancellation); fp32 states to 1e-6 absolute and increments to 2e-4 relative plus the
explicit fp32 state-rounding floor 4 eps |y| / |increment| (tests/util.py: inc_tol_f32 -- the state is O(1), an
increment O(1e-4..1e-2), and the state is rounded to fp32 every substep).
"""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import MOB, MU, inc_tol_f32, rel_l2

pytestmark = pytest.mark.gpu

# PDEOPT_FUZZ_SCALE=k multiplies the number of seeded cases (a longer soak run; the defaults keep the suite short)
import os  # noqa: E402

_SCALE = max(1, int(os.environ.get("PDEOPT_FUZZ_SCALE", "1")))

# tile-divisible, ragged (multiple of the 16-byte vector but not of the tile), vector-misaligned, tiny and 1-D
SHAPES = [(64, 128), (32, 256), (128, 128), (48, 40), (100, 100), (36, 24), (8, 8), (33, 20), (30, 7), (17, 64),
          (96, 64), (256, 1), (1, 96), (16, 132), (72, 200)]
EXPLICIT = {"euler": (L.INT_EULER, O.euler_step), "rk4": (L.INT_RK4, O.rk4_step),
            "tsit5": (L.INT_TSIT5, lambda f, t, y, dt: O.tsit5_step(f, t, y, dt)[0])}


def _check(got, want, y0, dtype, what):
    got, want, y0 = (np.asarray(a, np.float64) for a in (got, want, y0))
    assert np.isfinite(got).all(), what
    inc_g, inc_w = got - y0, want - y0
    if dtype is np.float64:
        assert rel_l2(inc_g, inc_w) < 1e-9, (what, rel_l2(inc_g, inc_w))
    else:
        assert np.max(np.abs(got - want)) < 1e-6 * max(1.0, np.max(np.abs(want))), (what, np.max(np.abs(got - want)))
        assert rel_l2(inc_g, inc_w) < inc_tol_f32(want, y0), (what, rel_l2(inc_g, inc_w), inc_tol_f32(want, y0))


@pytest.mark.parametrize("seed", range(64 * _SCALE))
def test_explicit_fd_random_configuration(seed):
    rng = np.random.default_rng(1000 + seed)
    dtype = [np.float32, np.float64][int(rng.integers(2))]
    nx, ny = SHAPES[int(rng.integers(len(SHAPES)))]
    batch = int(rng.integers(1, 7))
    group = [-1, 0, 1, 2, 3][int(rng.integers(5))]
    kind = ["ch", "ch", "ac", "ad"][int(rng.integers(4))]
    integ_name = ["euler", "rk4", "rk4", "tsit5"][int(rng.integers(4))]
    n = int(rng.integers(1, 8))
    h = float(rng.choice([0.01, 0.02, 0.013]))
    hy_scale = float(rng.choice([1.0, 1.0, 1.5]))  # anisotropic spacing now and then
    dom = P.Domain((nx, ny), ((-h * nx / 2, h * nx / 2), (0.0, h * hy_scale * ny)), "dimensionless")
    hx, hy = dom.dx
    kappas = 0.002 * (1.0 + 0.5 * rng.random(batch))
    per_env_kappa = bool(rng.integers(2)) and batch > 1
    if not per_env_kappa:
        kappas[:] = kappas[0]
    if kind == "ch":
        mu_name, mob_name = ["regsol", "cubic"][int(rng.integers(2))], ["c1mc", "one", "one_plus_sq"][int(rng.integers(3))]
        eq = P.CahnHilliard2DPeriodic(dom, float(kappas[0]), MU[mu_name], MOB[mob_name])
        rhs = lambda b: (lambda t, u: O.ch_rhs_fd(u, hx, hy, kappas[b], MU[mu_name], MOB[mob_name]))
        dt = 0.004 * min(hx, hy) ** 4 / 0.002  # well inside the explicit stability limit of the biharmonic
    elif kind == "ac":
        mu_name, mob_name = ["cubic", "regsol"][int(rng.integers(2))], ["one", "one_plus_sq", "const015"][int(rng.integers(3))]
        eq = P.AllenCahn2DPeriodic(dom, float(kappas[0]), MU[mu_name], MOB[mob_name])
        rhs = lambda b: (lambda t, u: O.ac_rhs_fd(u, hx, hy, kappas[b], MU[mu_name], MOB[mob_name]))
        dt = 0.05 * min(hx, hy) ** 2 / 0.002 * 0.1
    else:
        mu_name = "cubic"
        cx, cy = 0.1 * h * nx, 0.4 * h * hy_scale * ny

        def vel(t, xs, ys):
            r2 = ((xs - cx) ** 2 + (ys - cy) ** 2) / (2.0 * (4 * h) ** 2)
            return -0.1 * (xs - cx) / (4 * h) * np.exp(-r2), 0.07 * (ys - cy) / (4 * h) * np.exp(-r2)

        eq = P.AdvectionDiffusion2D(dom, vel, 0.1)
        vx, vy = eq.face_velocities(0.0)
        kappas[:] = 0.1
        per_env_kappa = False
        rhs = lambda b: (lambda t, u: O.ad_rhs_fd(u, hx, hy, vx, vy, 0.1))
        dt = 0.1 * min(hx, hy) ** 2 / 0.1
    if mu_name == "regsol":
        y0 = np.clip(0.5 + 0.1 * rng.standard_normal((batch, nx, ny)), 0.05, 0.95).astype(dtype)
    else:
        y0 = (0.1 * rng.standard_normal((batch, nx, ny))).astype(dtype)

    eng = P.HipEngine()
    eng.set_group_envs(group)
    opt = int(rng.integers(5))
    if opt == 1:
        eng.set_fuse_stages(-1)  # one kernel per stage
    elif opt == 2:
        eng.set_kernel_path(L.PATH_GENERIC)
    elif opt == 3:
        eng.set_graph(-1)  # no hipGraph replay
    elif opt == 4:
        eng.set_fuse_stages(1)  # stage pairs also where the single-pass kernel would run
    eng.configure(dtype=dtype, batch=batch, **eq._engine_problem())
    eq._engine_upload(eng, 0.0)
    if per_env_kappa:
        eng.set_env_params(0, kappa=[float(k) for k in kappas])
    eng.set_state(y0)
    integ, step = EXPLICIT[integ_name]
    # two calls: the second starts from device state left by the first (ping-pong buffers, graph replay)
    n1 = n // 2
    if n1:
        eng.advance(integ, dt, n1, 0.0)
    eng.advance(integ, dt, n - n1, n1 * dt)
    got = eng.get_state()
    what = dict(seed=seed, kind=kind, dtype=np.dtype(dtype).name, shape=(nx, ny), batch=batch, group=group, integ=integ_name,
                n=n, opt=opt, per_env_kappa=per_env_kappa, kernel=eng.last_kernel)
    eng.close()
    for b in range(batch):
        ref = y0[b].astype(np.float64)
        f = rhs(b)
        for i in range(n):
            ref = step(f, i * dt, ref, dt)
        _check(got[b], ref, y0[b], dtype, what)


@pytest.mark.parametrize("seed", range(16 * _SCALE))
def test_imex_random_configuration(seed):
    """IMEX through the LDS transforms (power-of-two sizes: both radix plans, 64 .. 1024) and through rocFFT
    (other sizes), random batch / groups / per-environment implicit operators"""
    rng = np.random.default_rng(2000 + seed)
    dtype = [np.float32, np.float64][int(rng.integers(2))]
    sizes = [64, 128, 256, 512, 1024, 48, 96, 80]
    nx = sizes[int(rng.integers(len(sizes)))]
    ny = sizes[int(rng.integers(len(sizes)))]
    while nx * ny > 1024 * 256:  # keep the fp64 numpy oracle in seconds
        ny = sizes[int(rng.integers(len(sizes)))]
    batch = int(rng.integers(1, 6))
    group = [-1, 0, 2, 3][int(rng.integers(4))]
    n = int(rng.integers(1, 5))
    h = 0.01
    dom = P.Domain((nx, ny), ((0.0, h * nx), (0.0, h * ny)), "dimensionless")
    hx, hy = dom.dx
    kappa, A, dt = 0.002, 0.5, 1e-6
    eq = P.CahnHilliard2DPeriodic(dom, kappa, MU["regsol"], MOB["c1mc"])
    solver = P.SemiImplicitFourierSpectral(A, eq.fourier_symbol, eq.fft, eq.ifft)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((batch, nx, ny)), 0.05, 0.95).astype(dtype)
    fused_size = all(s in (64, 128, 256, 512, 1024) for s in (nx, ny))
    per_env = bool(rng.integers(2)) and batch > 1 and fused_size
    kappas = np.full(batch, kappa)
    if per_env:
        kappas = kappa * (1.0 + rng.random(batch))

    eng = P.HipEngine()
    eng.set_group_envs(group)
    eng.configure(dtype=dtype, batch=batch, **eq._engine_problem())
    eq._engine_upload(eng, 0.0)
    solver.configure_engine(eng, eq)
    if per_env:
        eng.set_env_params(0, kappa=[float(k) for k in kappas])
        eng.set_env_imex_scale(0, kappas / kappa)
    eng.set_state(y0)
    eng.advance(L.INT_IMEX, dt, n)
    got = eng.get_state()
    what = dict(seed=seed, dtype=np.dtype(dtype).name, shape=(nx, ny), batch=batch, group=group, n=n, per_env=per_env,
                kernel=eng.last_kernel)
    assert ("imex_fused_lds_fft" in eng.last_kernel) == fused_size, what
    eng.close()
    for b in range(batch):
        sym = O.ch_fourier_symbol(nx, ny, hx, hy, kappas[b])
        f = lambda t, u: O.ch_rhs_fd(u, hx, hy, kappas[b], MU["regsol"], MOB["c1mc"])
        ref = y0[b].astype(np.float64)
        for i in range(n):
            ref = O.imex_step(f, i * dt, ref, dt, A, sym)
        _check(got[b], ref, y0[b], dtype, what)


@pytest.mark.parametrize("seed", range(16 * _SCALE))
def test_strang_random_configuration(seed):
    """Strang split steps of the GPE: LDS transforms / rocFFT sizes, real and imaginary time, per-environment
    interaction strengths and potentials, static and moving light spots"""
    rng = np.random.default_rng(3000 + seed)
    dtype = [np.float32, np.float64][int(rng.integers(2))]
    sizes = [64, 128, 256, 512, 48, 96]
    nx = sizes[int(rng.integers(len(sizes)))]
    ny = nx if rng.integers(2) else sizes[int(rng.integers(len(sizes)))]
    while nx * ny > 512 * 256:
        ny = sizes[int(rng.integers(len(sizes)))]
    batch = int(rng.integers(1, 5))
    group = [-1, 0, 1, 3][int(rng.integers(4))]
    n = int(rng.integers(1, 5))
    tscale = [1.0, -1j][int(rng.integers(2))]
    kinetic = bool(rng.integers(2))
    moving = bool(rng.integers(2))
    Lb = 24.0
    # the split step takes ONE spacing (solvers.py:117 normalises with dx[0]^2): square cells
    dom = P.Domain((nx, ny), ((-Lb / 2, Lb / 2), (-Lb / 2 * ny / nx, Lb / 2 * ny / nx)), "dimensionless")
    if moving:
        lights = lambda t, x, y: 20.0 * (1.0 + 50.0 * t) * np.exp(-((x - 1.0 - 300.0 * t) ** 2 + (y + 0.5) ** 2) / 3.0)
    else:
        lights = lambda t, x, y: 0.05 * x - 0.02 * y
    k0, e, dt = 400.0, 0.2, 1e-3
    eq = P.GPE2DTSControl(dom, k0, e, lights, trap_factor=1.0, kinetic=kinetic)
    X, Y = dom.mesh()
    states = []
    for b in range(batch):
        psi = np.exp(-(X**2 / (14.0 + b) + Y**2 / 10.0)) * np.exp(0.3j * X - 0.1j * b * Y)
        psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * dom.dx[0] ** 2)
        states.append(np.stack([psi.real, psi.imag], axis=-1))
    y0 = np.stack(states).astype(dtype)
    ks = k0 * (1.0 + 0.1 * np.arange(batch))
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, tscale)
    sol = P.diffeqsolve(eq, solver, 0.0, n * dt, dt, y0, engine=_strang_engine(group, ks))
    got = sol.ys[-1]
    what = dict(seed=seed, dtype=np.dtype(dtype).name, shape=(nx, ny), batch=batch, group=group, n=n, tscale=tscale,
                kinetic=kinetic, moving=moving)
    A_term = np.asarray(eq.A_term)
    for b in range(batch):
        bt = lambda t, y: O.gpe_b_terms(y, X, Y, ks[b], e, 1.0, lights(t, X, Y))
        ref = y0[b].astype(np.float64)
        for i in range(n):
            ref = O.strang_step(bt, i * dt, ref, dt, A_term, dom.dx[0], tscale)
        want = ref
        if dtype is np.float64:
            assert rel_l2(got[b], want) < 1e-10, (what, rel_l2(got[b], want))
        else:
            assert rel_l2(got[b], want) < 2e-5, (what, rel_l2(got[b], want))


def _strang_engine(group, ks):
    class _Eng(P.HipEngine):
        """engine with the group size and per-environment interaction strengths applied at configure time"""

        def configure(self, **kw):
            super().configure(**kw)
            self.set_env_gpe_k(0, ks)

    eng = _Eng()
    eng.set_group_envs(group)
    return eng
