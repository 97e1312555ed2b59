"""GPU parity tests: the HIP path (through the Python shims -> ctypes -> C ABI) against
  (1) the goldens generated from the reference's own source (tests/golden, oracle/gen_golden.py),
  (2) the CPU oracle on seeded inputs at sizes it finishes in seconds,
  (3) the reference's own known-answer tests (sympy slope, tanh profile, Thomas-Fermi).
Tolerances are stated in tests/util.py."""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import MOB, MU, TOL, inc_tol_f32, rel_l2, std_domain, white_noise_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    return P.engine.default_engine()


def _force(engine, path):
    engine.set_kernel_path(path)


# ------------------------------------------------------------------ RHS vs reference goldens
@pytest.mark.parametrize("kind", ["ch_fd", "ac_fd"])
@pytest.mark.parametrize("path", [L.PATH_GENERIC, L.PATH_AUTO])
def test_rhs_against_reference_goldens(golden, engine, kind, path):
    z = golden("rhs_cases.npz")
    keys = sorted(k[:-4] for k in z.files if k.startswith(kind + "/") and k.endswith("/rhs"))
    assert len(keys) >= 20
    _force(engine, path)
    try:
        used = set()
        for key in keys:
            _, mu, mob, tag = key.split("/")
            nx, ny = (int(v) for v in tag.split("_")[0].split("x"))
            dom = std_domain(P, nx, ny)
            cls = P.CahnHilliard2DPeriodic if kind == "ch_fd" else P.AllenCahn2DPeriodic
            eq = cls(dom, 0.002, MU[mu], MOB[mob])
            u, want = z[key + "/u"], z[key + "/rhs"]
            got = eq.rhs(u, 0.0)
            used.add(engine.last_kernel)
            assert got.dtype == want.dtype and got.shape == want.shape
            assert rel_l2(got, want) < TOL[want.dtype], (key, engine.last_kernel, rel_l2(got, want))
        if path == L.PATH_AUTO:
            assert any("tiled" in k for k in used), used  # the 128x128 cases take the LDS-tiled path
        else:
            assert all("generic" in k for k in used), used
    finally:
        _force(engine, L.PATH_AUTO)


# ------------------------------------------------------------------ RHS vs oracle, tiled shapes
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("eqname", ["ch", "ac"])
@pytest.mark.parametrize(
    "mu,mob,kind",
    [("cubic", "one_plus_sq", "sym"), ("regsol", "c1mc", "c"), ("legendre", "explegendre", "c")],
)
def test_rhs_tiled_vs_oracle(engine, dtype, eqname, mu, mob, kind):
    rng = np.random.default_rng(7)
    leg_mu = P.ChemicalPotentialLegendrePolynomials([0.3, 0.1, -0.2, -0.1, 0.45], prior_fn=lambda c: np.log(c / (1 - c)))
    leg_d = P.DiffusionLegendrePolynomials([0.2, -0.1, 0.05, -0.02])
    mu_fn = leg_mu if mu == "legendre" else MU[mu]
    mob_fn = leg_d if mob == "explegendre" else MOB[mob]
    for (nx, ny, batch) in ((64, 128, 3), (32, 256, 1), (96, 128, 2)):
        dom = std_domain(P, nx, ny)
        cls = P.CahnHilliard2DPeriodic if eqname == "ch" else P.AllenCahn2DPeriodic
        eq = cls(dom, 0.002, mu_fn, mob_fn)
        u = white_noise_state(rng, (batch, nx, ny), dtype, kind)
        got = eq.rhs(u, 0.0)
        assert "tiled" in engine.last_kernel, engine.last_kernel
        hx, hy = dom.dx
        fn = O.ch_rhs_fd if eqname == "ch" else O.ac_rhs_fd
        for b in range(batch):
            want = fn(u[b], hx, hy, 0.002, mu_fn, mob_fn)
            assert rel_l2(got[b], want) < TOL[np.dtype(dtype)], (nx, ny, b, rel_l2(got[b], want))
            want64 = fn(u[b].astype(np.float64), hx, hy, 0.002, mu_fn, mob_fn)
            assert rel_l2(got[b], want64) < 2.5 * TOL[np.dtype(dtype)]
        # tiled and generic kernels agree with each other
        _force(engine, L.PATH_GENERIC)
        try:
            gen = eq.rhs(u, 0.0)
        finally:
            _force(engine, L.PATH_AUTO)
        assert rel_l2(got, gen) < TOL[np.dtype(dtype)]


def test_manufactured_solution_convergence():
    """tests/test_rhs_convergence.py:14-77 re-expressed on the build's API: slope 2.0 +- 10 %."""
    import sympy as sp
    from sympy.utilities.lambdify import lambdify

    x, y, t = sp.symbols("x y t", real=True)
    u = sp.sin(2 * x) * sp.cos(3 * y) * sp.exp(-0.7 * t)
    kappa = 1e-2
    mu = u**3 - u - kappa * (sp.diff(u, x, 2) + sp.diff(u, y, 2))
    D = 1 + u**2
    exprs = {
        "ch": sp.diff(D * sp.diff(mu, x), x) + sp.diff(D * sp.diff(mu, y), y),
        "ac": -D * mu,
    }
    u_fn = lambdify((x, y, t), u, "numpy")
    for name, cls in (("ch", P.CahnHilliard2DPeriodic), ("ac", P.AllenCahn2DPeriodic)):
        ex_fn = lambdify((x, y, t), exprs[name], "numpy")
        hs, errs = [], []
        for n in (32, 64, 128, 256, 512):
            Lb = 2 * np.pi
            dom = P.Domain((n, n), ((-Lb / 2, Lb / 2), (-Lb / 2, Lb / 2)), "dimensionless")
            X, Y = dom.mesh()
            eq = cls(dom, kappa, lambda c: c**3 - c, lambda c: 1 + c**2, derivs="fd")
            got = eq.rhs(u_fn(X, Y, 0.0), 0)
            exact = ex_fn(X, Y, 0.0)
            errs.append(np.sqrt(np.sum((got - exact) ** 2)) / np.sqrt(np.sum(exact**2)))
            hs.append(dom.dx[0])
        slope = np.polyfit(np.log(hs), np.log(errs), 1)[0]
        np.testing.assert_allclose(slope, 2.0, rtol=0.1)


# ------------------------------------------------------------------ explicit integrators
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(64, 128), (48, 40), (256, 1)])
@pytest.mark.parametrize("solver", ["euler", "rk4"])
def test_explicit_trajectory_vs_oracle(engine, dtype, shape, solver):
    rng = np.random.default_rng(3)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    mu_fn, mob_fn = MU["regsol"], MOB["c1mc"]
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, mu_fn, mob_fn)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((2, nx, ny)), 0.05, 0.95).astype(dtype)
    dt, n = 2e-7, 8
    s = P.Euler() if solver == "euler" else P.RK4()
    sol = P.diffeqsolve(eq, s, t0=0.0, t1=n * dt, dt0=dt, y0=y0)
    hx, hy = dom.dx
    f = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, mu_fn, mob_fn)
    step = O.euler_step if solver == "euler" else O.rk4_step
    for b in range(2):
        ref = y0[b].astype(np.float64)
        for i in range(n):
            ref = step(f, i * dt, ref, dt)
        inc_ref = ref - y0[b].astype(np.float64)
        inc_got = sol.ys[-1][b].astype(np.float64) - y0[b].astype(np.float64)
        # fp32: the state is O(1) and the increment O(1e-3): state rounding (6e-8) bounds the
        # increment's relative accuracy at ~1e-4
        tol = 1e-10 if dtype is np.float64 else 5e-4
        assert rel_l2(inc_got, inc_ref) < tol, (shape, solver, rel_l2(inc_got, inc_ref))
        assert np.max(np.abs(sol.ys[-1][b] - ref)) < (1e-13 if dtype is np.float64 else 5e-7)


def test_allen_cahn_rk4_config2_slice(engine):
    """BASELINE config 2 (AC 512^2 fp32 RK4, dt 5e-5) on 4 envs for 20 substeps vs the oracle."""
    rng = np.random.default_rng(11)
    nx = ny = 512
    dom = std_domain(P, nx, ny)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    y0 = (0.01 * rng.standard_normal((4, nx, ny))).astype(np.float32)
    dt, n = 5e-5, 20
    sol = P.diffeqsolve(eq, P.RK4(), 0.0, n * dt, dt, y0)
    assert "rk4_quad" in sol.stats["kernel"]  # the whole RK4 substep in one pass for Allen-Cahn fp32
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
    ref = y0[1].astype(np.float64)
    for i in range(n):
        ref = O.rk4_step(f, 0.0, ref, dt)
    assert rel_l2(sol.ys[-1][1], ref) < 5e-6


def test_batch_equals_single_bitwise(engine):
    """an environment's trajectory does not depend on which batch it rides in"""
    rng = np.random.default_rng(5)
    dom = std_domain(P, 64, 128)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((5, 64, 128)), 0.05, 0.95).astype(np.float32)
    batched = P.diffeqsolve(eq, P.RK4(), 0.0, 2e-6, 2e-7, y0).ys[-1]
    for b in (0, 3):
        single = P.diffeqsolve(eq, P.RK4(), 0.0, 2e-6, 2e-7, y0[b]).ys[-1]
        np.testing.assert_array_equal(single, batched[b])
    again = P.diffeqsolve(eq, P.RK4(), 0.0, 2e-6, 2e-7, y0).ys[-1]
    np.testing.assert_array_equal(again, batched)  # run-to-run determinism


def test_mass_conservation_full_size(engine):
    """size-independent property at BASELINE's full grid: CH conserves the mean (flux form)."""
    rng = np.random.default_rng(1)
    dom = std_domain(P, 1024, 1024)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((2, 1024, 1024)), 0.05, 0.95).astype(np.float32)
    sol = P.diffeqsolve(eq, P.RK4(), 0.0, 100 * 2e-7, 2e-7, y0)
    y1 = sol.ys[-1]
    assert np.all(np.isfinite(y1))
    for b in range(2):
        assert abs(y1[b].astype(np.float64).mean() - y0[b].astype(np.float64).mean()) < 2e-7
    assert np.linalg.norm(y1 - y0) > 0


def test_per_env_parameters(engine):
    """control parameters travel with the environment: per-env kappa == separate solves"""
    rng = np.random.default_rng(9)
    dom = std_domain(P, 64, 128)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((3, 64, 128)), 0.05, 0.95)
    kappas = [0.001, 0.002, 0.004]
    eq = P.CahnHilliard2DPeriodic(dom, kappas[0], MU["regsol"], MOB["c1mc"])
    eng = P.HipEngine()
    eng.configure(dtype=np.float64, batch=3, **eq._engine_problem())
    eng.set_env_params(0, kappa=kappas)
    eng.set_state(y0)
    eng.advance(L.INT_RK4, 2e-7, 5)
    got = eng.get_state()
    for b, k in enumerate(kappas):
        e = P.CahnHilliard2DPeriodic(dom, k, MU["regsol"], MOB["c1mc"])
        want = P.diffeqsolve(e, P.RK4(), 0.0, 1e-6, 2e-7, y0[b]).ys[-1]
        np.testing.assert_array_equal(got[b], want)
    eng.close()


def test_saveat_linear_interpolation(engine):
    rng = np.random.default_rng(2)
    dom = std_domain(P, 32, 32)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    y0 = 0.1 * rng.standard_normal((32, 32))
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
    ts = [0.0, 1.3e-4, 2e-4, 4.9e-4, 5e-4]
    model = P.PDEModel(P.AllenCahn2DPeriodic, dom, P.Euler)
    ys = model.solve(dict(kappa=0.002, mu=MU["cubic"], R=MOB["one"]), y0, ts, dt0=1e-4)
    want = O.solve_saveat(lambda t, y, dt: O.euler_step(f, t, y, dt), y0, ts, 1e-4)
    assert ys.shape == (5, 32, 32)
    np.testing.assert_allclose(ys, want, rtol=0, atol=1e-13)


# ------------------------------------------------------------------ spectral integrators
def test_imex_trajectory_vs_reference_golden(golden):
    z = golden("trajectories.npz")
    dom = P.Domain((64, 64), ((-0.32, 0.32), (-0.32, 0.32)), "dimensionless")
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
    dt = float(z["imex/dt"])
    model_ts = [i * dt for i in range(11)]
    sol = P.diffeqsolve(eq, solver, 0.0, 10 * dt, dt, z["imex/y0"], saveat=P.SaveAt(ts=model_ts))
    inc_ref = z["imex/ys"][-1] - z["imex/y0"]
    assert rel_l2(sol.ys[-1] - z["imex/y0"], inc_ref) < 1e-9
    np.testing.assert_allclose(sol.ys[1:], z["imex/ys"], rtol=0, atol=1e-13)


def test_imex_1d_reference_golden(golden):
    z = golden("trajectories.npz")
    dom = P.Domain((256, 1), ((-1.28, 1.28), (-0.005, 0.005)), "dimensionless")
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
    sol = P.diffeqsolve(eq, solver, 0.0, 200 * 5e-5, 5e-5, z["imex1d/y0"])
    np.testing.assert_allclose(sol.ys[-1], z["imex1d/y200"], rtol=0, atol=1e-10)


@pytest.mark.parametrize("name,tscale,kinetic", [("zeroA_imag", -1j, False), ("realA_real", 1.0, True), ("realA_imag", -1j, True)])
def test_strang_trajectory_vs_reference_golden(golden, name, tscale, kinetic):
    z = golden("trajectories.npz")
    dom = P.Domain((48, 48), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    eq = P.GPE2DTSControl(dom, 1000.0, 0.1, lambda t, x, y: 0.05 * x, trap_factor=1.0, kinetic=kinetic)
    if kinetic:
        np.testing.assert_allclose(eq.A_term, z["strang/A_real"], rtol=1e-14)
    np.testing.assert_allclose(eq.B_terms(z["strang/y0"], 0.0), z["strang/b_terms"], rtol=1e-13, atol=1e-13)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, tscale)
    ts = [i * 1e-3 for i in range(6)]
    sol = P.diffeqsolve(eq, solver, 0.0, 5e-3, 1e-3, z["strang/y0"], saveat=P.SaveAt(ts=ts))
    np.testing.assert_allclose(sol.ys[1:], z[f"strang/{name}/ys"], rtol=0, atol=2e-12)
    # fp32 batch of 3 copies
    y32 = np.repeat(z["strang/y0"][None].astype(np.float32), 3, axis=0)
    sol32 = P.diffeqsolve(eq, solver, 0.0, 5e-3, 1e-3, y32)
    for b in range(3):
        assert rel_l2(sol32.ys[-1][b], z[f"strang/{name}/ys"][-1]) < 2e-5


# ------------------------------------------------------------------ reference known-answer tests
def test_1d_cahn_hilliard_tanh():
    """tests/test_solvers.py:21-61 (and its PDEModel twin :208-251): IMEX -> tanh(x/sqrt(2 kappa))."""
    nx, ny = 256, 1
    dom = std_domain(P, nx, ny)
    kappa = 0.002
    u0 = np.ones((nx, ny))
    u0[: nx // 2, :] = -1.0
    model = P.PDEModel(P.CahnHilliard2DPeriodic, dom, P.SemiImplicitFourierSpectral)
    ts = np.linspace(0.0, 10.0, 200)
    ys = model.solve(
        dict(kappa=kappa, mu=lambda c: c**3 - c, D=lambda c: np.ones_like(c), derivs="fd"),
        u0, ts, solver_parameters={"A": 0.5}, dt0=0.00005,
    )
    assert ys.shape == (200, nx, ny)
    analytic = np.tanh(dom.axes()[0] / np.sqrt(2 * kappa))
    np.testing.assert_allclose(ys[-1].squeeze()[nx // 4: 3 * nx // 4], analytic[nx // 4: 3 * nx // 4], rtol=1e-3, atol=1e-3)


def test_1d_allen_cahn_tanh_tsit5_pid():
    """tests/test_solvers.py:64-104: Tsit5 + PIDController(rtol=1e-4, atol=1e-6) -> tanh."""
    nx, ny = 256, 1
    dom = std_domain(P, nx, ny)
    kappa = 0.002
    eq = P.AllenCahn2DPeriodic(dom, kappa, lambda c: c**3 - c, lambda c: np.ones_like(c), derivs="fd")
    u0 = np.ones((nx, ny))
    u0[: nx // 2, :] = -1.0
    sol = P.diffeqsolve(
        eq, P.Tsit5(), t0=0.0, t1=10.0, dt0=0.00005, y0=u0,
        saveat=P.SaveAt(ts=np.linspace(0.0, 10.0, 200)),
        stepsize_controller=P.PIDController(rtol=1e-4, atol=1e-6), max_steps=1000000,
    )
    analytic = np.tanh(dom.axes()[0] / np.sqrt(2 * kappa))
    np.testing.assert_allclose(sol.ys[-1].squeeze()[nx // 4: 3 * nx // 4], analytic[nx // 4: 3 * nx // 4], rtol=1e-3, atol=1e-3)
    assert sol.stats["num_accepted_steps"] > 10


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tsit5_dense_output_vs_oracle(dtype):
    """pdeopt_tsit5_dense: the 4th-order interpolant of diffrax.Tsit5 at SaveAt points inside a step (several per
    step), fixed and adaptive stepping, batched; against the oracle's restatement of Tsitouras' section 4"""
    rng = np.random.default_rng(14)
    dom = std_domain(P, 64, 128)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((2, 64, 128)), 0.05, 0.95).astype(dtype)
    hx, hy = dom.dx
    f = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, MU["regsol"], MOB["c1mc"])
    dt = 2e-7
    ts = [0.0, 0.2 * dt, 0.7 * dt, 1.6 * dt, 1.9 * dt, 2 * dt, 3 * dt]
    sol = P.diffeqsolve(eq, P.Tsit5(), ts[0], ts[-1], dt, y0, saveat=P.SaveAt(ts=ts))
    for b in range(2):
        y, out = y0[b].astype(np.float64), [y0[b].astype(np.float64)]
        for i, inner in enumerate(([0.2, 0.7], [0.6, 0.9], [])):
            y1, _, _, ks = O.tsit5_step(f, i * dt, y, dt, return_slopes=True)
            out += [O.tsit5_dense(y, dt, ks, th) for th in inner]
            y = y1
            if i >= 1:
                out.append(y)
        want = np.stack(out)
        inc, inc_ref = sol.ys[:, b].astype(np.float64) - y0[b], want - y0[b]
        assert rel_l2(inc[1:], inc_ref[1:]) < (1e-10 if dtype is np.float64 else 5e-3), rel_l2(inc[1:], inc_ref[1:])
        assert np.max(np.abs(sol.ys[:, b] - want)) < (1e-13 if dtype is np.float64 else 5e-7)


@pytest.mark.parametrize("shape", [(48, 1), (64, 128)])
def test_per_environment_step_sizes_equal_solo_solves(shape):
    """pdeopt_tsit5_trial_env / commit_env: each environment of a batch steps as it would alone (generic and
    LDS-tiled stage kernels); slopes are stored scaled by dt_b / dt_ref, so results agree to rounding"""
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
    u = np.ones((nx, ny))
    u[: nx // 2] = -1.0
    rng = np.random.default_rng(9)
    y0 = np.stack([u, 0.05 * rng.standard_normal((nx, ny)), 0.9 * u + 0.3 * rng.standard_normal((nx, ny))])
    ts = [0.0, 0.013, 0.05, 0.2]
    ctl = dict(rtol=1e-5, atol=1e-7)
    eng = P.HipEngine()
    solo = [P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0[b], saveat=P.SaveAt(ts=ts),
                          stepsize_controller=P.PIDController(**ctl), engine=eng) for b in range(3)]
    both = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0, saveat=P.SaveAt(ts=ts),
                         stepsize_controller=P.PIDController(**ctl, per_environment=True), engine=eng)
    for b in range(3):
        np.testing.assert_allclose(both.ys[:, b], solo[b].ys, rtol=0, atol=1e-10)
        assert abs(both.stats["num_accepted_steps"][b] - solo[b].stats["num_accepted_steps"]) <= 1
    assert len(set(both.stats["num_accepted_steps"])) > 1
    # a uniform trial after per-environment ones recomputes the FSAL slope (it carried per-environment scales)
    again = P.diffeqsolve(eq, P.Tsit5(), 0.0, 0.2, 1e-4, y0[1], saveat=P.SaveAt(ts=ts),
                          stepsize_controller=P.PIDController(**ctl), engine=eng)
    np.testing.assert_array_equal(again.ys, solo[1].ys)
    eng.close()


def test_tsit5_fixed_step_vs_oracle():
    rng = np.random.default_rng(4)
    dom = std_domain(P, 32, 32)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one_plus_sq"])
    y0 = 0.1 * rng.standard_normal((32, 32))
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one_plus_sq"])
    sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, 5e-4, 1e-4, y0)
    ref = y0
    for i in range(5):
        ref, _, _ = O.tsit5_step(f, 0.0, ref, 1e-4)
    np.testing.assert_allclose(sol.ys[-1], ref, rtol=0, atol=1e-13)


def test_2d_gross_pitaevskii_thomas_fermi():
    """tests/test_solvers.py:107-205: imaginary-time Strang -> Thomas-Fermi density."""
    atoms = 5e5
    hbar = 1.05e-34
    omega = 2 * np.pi * 10
    omega_z = np.sqrt(8) * omega
    mass = 3.8175406e-26
    a0 = 5.29177210903e-11
    a_s = 100 * a0
    N = 128
    x_s = np.sqrt(hbar / (mass * omega))
    t_s = 1 / omega
    Lx_ = 150e-6 / x_s
    k = 4 * np.pi * a_s * atoms * np.sqrt((mass * omega_z) / (2 * np.pi * hbar))
    t_final_ = 0.1 / t_s
    dt_ = 1e-5 / t_s
    dom = P.Domain((N, N), ((-Lx_ / 2, Lx_ / 2), (-Lx_ / 2, Lx_ / 2)), "dimensionless")
    # initialize_Psi(N, width=100, vortexnumber=0): numerics/utils/initialization_utils.py:11-34
    ii = np.arange(N) - N // 2
    psi0 = np.exp(-((ii[:, None] / 100.0) ** 2) - (ii[None, :] / 100.0) ** 2).astype(complex) * x_s
    psi0 /= np.sqrt(np.sum(np.abs(psi0) ** 2) * dom.dx[0] ** 2)
    eq = P.GPE2DTSControl(dom, k, 0.0, lambda a, b, c: 0.0, trap_factor=1.0)
    solver = P.StrangSplitting(eq.A_term, eq.domain.dx[0], eq.fft, eq.ifft, -1j)
    sol = P.diffeqsolve(
        eq, solver, t0=0.0, t1=t_final_, dt0=dt_, y0=np.stack([psi0.real, psi0.imag], axis=-1),
        saveat=P.SaveAt(ts=np.linspace(0.0, t_final_, 100)), max_steps=1000000,
    )
    X, Y = dom.mesh()
    g = k
    mu_tf = np.sqrt((1.0 * g * np.sqrt(0.5) * np.sqrt(0.5)) / (2.0 * np.pi))
    V = 0.5 * (0.5 * X**2 + 0.5 * Y**2)
    n = np.clip((mu_tf - V) / g, 0.0, None)
    n = n * (1.0 / (np.sum(n) * (X[1, 0] - X[0, 0]) * (Y[0, 1] - Y[0, 0]) + 1e-12))
    dens = sol.ys[-1][..., 0] ** 2 + sol.ys[-1][..., 1] ** 2
    np.testing.assert_allclose(n, dens, rtol=1e-3, atol=1e-3)


# ------------------------------------------------------------------ advection-diffusion (unpinned)
def test_advection_diffusion_vs_oracle_and_conservation():
    rng = np.random.default_rng(6)
    nx = ny = 128
    dom = std_domain(P, nx, ny, h=0.02)

    def vel(t, xs, ys):
        r2 = ((xs - 0.4) ** 2 + (ys - 0.4) ** 2) / (2.0 * 0.01)
        return -0.1 * (xs - 0.4) / 0.01 * np.exp(-r2), -0.1 * (ys - 0.4) / 0.01 * np.exp(-r2)

    eq = P.AdvectionDiffusion2D(dom, vel, 0.1)
    u0 = 0.5 + 0.01 * rng.standard_normal((nx, ny))
    hx, hy = dom.dx
    vx, vy = eq.face_velocities(0.0)
    np.testing.assert_allclose(eq.rhs(u0, 0.0), O.ad_rhs_fd(u0, hx, hy, vx, vy, 0.1), rtol=0, atol=1e-9)
    sol = P.diffeqsolve(eq, P.Euler(), 0.0, 0.05, 1e-4, u0)  # config 1: 500 Euler substeps
    ref = O.integrate(lambda t, y, dt: O.euler_step(lambda tt, u: O.ad_rhs_fd(u, hx, hy, vx, vy, 0.1), t, y, dt), u0, 0.0, 0.05, 1e-4)
    np.testing.assert_allclose(sol.ys[-1], ref, rtol=0, atol=1e-12)
    assert abs(sol.ys[-1].mean() - u0.mean()) < 1e-14
    assert sol.stats["num_steps"] == 500


# ------------------------------------------------------------------ reductions
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reductions(dtype):
    rng = np.random.default_rng(8)
    dom = std_domain(P, 96, 80)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    eng = P.HipEngine()
    y = (0.5 + 0.01 * rng.standard_normal((4, 96, 80))).astype(dtype)
    eng.configure(dtype=dtype, batch=4, **eq._engine_problem())
    eng.set_state(y)
    y64 = y.astype(np.float64)
    np.testing.assert_allclose(eng.reduce(L.RED_MEAN), y64.mean(axis=(1, 2)), rtol=1e-13)
    np.testing.assert_allclose(eng.reduce(L.RED_VAR), y64.var(axis=(1, 2)), rtol=1e-11)
    np.testing.assert_array_equal(eng.reduce(L.RED_MIN), y64.min(axis=(1, 2)))
    np.testing.assert_array_equal(eng.reduce(L.RED_MAX), y64.max(axis=(1, 2)))
    np.testing.assert_array_equal(eng.reduce(L.RED_NONFINITE), np.zeros(4))
    y[2, 5, 5] = np.nan
    eng.set_state(y)
    np.testing.assert_array_equal(eng.reduce(L.RED_NONFINITE), [0, 0, 1, 0])
    eng.close()


def test_error_codes():
    eng = P.HipEngine()
    with pytest.raises(P.PdeoptError):
        eng.advance(L.INT_RK4, 1e-3, 1)  # not configured -> ESTATE
    dom = std_domain(P, 32, 32)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    eng.configure(dtype=np.float32, batch=1, **eq._engine_problem())
    with pytest.raises(ValueError):
        eng.advance(L.INT_STRANG, 1e-3, 1)  # Strang on CH -> EINVAL -> ValueError
    with pytest.raises(ValueError):
        eng.advance(99, 1e-3, 1)
    with pytest.raises(P.PdeoptError):
        eng.advance(L.INT_IMEX, 1e-3, 1)  # no symbol uploaded -> ESTATE
    with pytest.raises(ValueError):
        eng.set_state(np.zeros((1, 16, 16), np.float32))
    eng.close()


# ------------------------------------------------------------------ pseudo-spectral RHS
def test_rhs_fourier_against_reference_goldens(golden, engine):
    z = golden("rhs_cases.npz")
    n = 0
    for kind, cls in (("ch_fourier", P.CahnHilliard2DPeriodic), ("ac_fourier", P.AllenCahn2DPeriodic)):
        for key in sorted(k[:-4] for k in z.files if k.startswith(kind + "/") and k.endswith("/rhs")):
            _, mu, mob, tag = key.split("/")
            nx, ny = (int(v) for v in tag.split("_")[0].split("x"))
            eq = cls(std_domain(P, nx, ny), 0.002, MU[mu], MOB[mob], derivs="fourier")
            got = eq.rhs(z[key + "/u"], 0.0)
            assert "rhs_fourier" in engine.last_kernel
            # 7 (CH) / 3 (AC) FFT round trips with the (2 pi k)^4 ~ 1e10 amplification of the biharmonic
            assert rel_l2(got, z[key + "/rhs"]) < 1e-9, (key, rel_l2(got, z[key + "/rhs"]))
            n += 1
    assert n >= 4


def test_rhs_fourier_spectral_accuracy_and_batch():
    """spectral RHS of the manufactured solution is exact to rounding (reference: 6.8e-10 at N=512)"""
    import sympy as sp
    from sympy.utilities.lambdify import lambdify

    x, y, t = sp.symbols("x y t", real=True)
    u = sp.sin(2 * x) * sp.cos(3 * y)
    kappa = 1e-2
    mu = u**3 - u - kappa * (sp.diff(u, x, 2) + sp.diff(u, y, 2))
    D = 1 + u**2
    ch = lambdify((x, y), sp.diff(D * sp.diff(mu, x), x) + sp.diff(D * sp.diff(mu, y), y), "numpy")
    ac = lambdify((x, y), -D * mu, "numpy")
    Lb = 2 * np.pi
    dom = P.Domain((64, 64), ((-Lb / 2, Lb / 2), (-Lb / 2, Lb / 2)), "dimensionless")
    X, Y = dom.mesh()
    ue = lambdify((x, y), u, "numpy")(X, Y)
    for cls, exact in ((P.CahnHilliard2DPeriodic, ch(X, Y)), (P.AllenCahn2DPeriodic, ac(X, Y))):
        eq = cls(dom, kappa, lambda c: c**3 - c, lambda c: 1 + c**2, derivs="fourier")
        got = eq.rhs(np.stack([ue, 0.5 * ue]), 0)  # batch of two
        assert rel_l2(got[0], exact) < 1e-11
        want1 = (O.ch_rhs_fourier if cls is P.CahnHilliard2DPeriodic else O.ac_rhs_fourier)(
            0.5 * ue, dom.dx[0], dom.dx[1], kappa, lambda c: c**3 - c, lambda c: 1 + c**2)
        assert rel_l2(got[1], want1) < 1e-11
    # fp32 works too (tolerance of an fp32 FFT chain)
    eq = P.AllenCahn2DPeriodic(dom, kappa, lambda c: c**3 - c, lambda c: 1 + c**2, derivs="fourier")
    assert rel_l2(eq.rhs(ue.astype(np.float32), 0), ac(X, Y)) < 2e-5


def test_explicit_step_with_fourier_rhs():
    rng = np.random.default_rng(12)
    dom = std_domain(P, 32, 48)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one_plus_sq"], derivs="fourier")
    y0 = 0.1 * rng.standard_normal((32, 48))
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fourier(u, hx, hy, 0.002, MU["cubic"], MOB["one_plus_sq"])
    got = P.diffeqsolve(eq, P.RK4(), 0.0, 4e-4, 1e-4, y0).ys[-1]
    ref = y0
    for _ in range(4):
        ref = O.rk4_step(f, 0.0, ref, 1e-4)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13)
    with pytest.raises(ValueError, match="Invalid derivative type"):
        P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"], derivs="spectral")


# ------------------------------------------------------------------ API corners
def test_per_env_closure_coefficients():
    """closure coefficient VALUES travel with the environment (same structure across the batch)"""
    rng = np.random.default_rng(21)
    dom = std_domain(P, 64, 128)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((3, 64, 128)), 0.05, 0.95).astype(np.float32)
    omegas = [2.5, 3.0, 3.5]
    mobs = [0.5, 1.0, 2.0]
    mk = lambda om, m: P.CahnHilliard2DPeriodic(
        dom, 0.002, P.polynomial(om, -2 * om, logit_prior=True), P.polynomial(0.0, m, -m))
    eqs = [mk(om, m) for om, m in zip(omegas, mobs)]
    eng = P.HipEngine()
    eng.configure(dtype=np.float32, batch=3, **eqs[0]._engine_problem())
    eng.set_env_params(0, mu_coef=[e._mu_desc.coef for e in eqs], mob_coef=[e._mob_desc.coef for e in eqs])
    eng.set_state(y0)
    eng.advance(L.INT_RK4, 2e-7, 4)
    got = eng.get_state()
    for b, e in enumerate(eqs):
        want = P.diffeqsolve(e, P.RK4(), 0.0, 8e-7, 2e-7, y0[b]).ys[-1]
        np.testing.assert_array_equal(got[b], want)
    ptr, nbytes = eng.state_device_ptr()
    assert ptr and nbytes == 3 * 64 * 128 * 4
    eng.close()


def test_saveat_variants_and_max_steps():
    rng = np.random.default_rng(22)
    dom = std_domain(P, 32, 32)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    y0 = 0.1 * rng.standard_normal((32, 32))
    sol = P.diffeqsolve(eq, P.Euler(), 0.0, 1e-3, 1e-4, y0, saveat=P.SaveAt(t0=True, t1=True))
    assert sol.ys.shape == (2, 32, 32) and sol.stats["num_steps"] == 10
    np.testing.assert_array_equal(sol.ys[0], y0)
    with pytest.raises(RuntimeError, match="max_steps"):
        P.diffeqsolve(eq, P.Euler(), 0.0, 1e-3, 1e-4, y0, max_steps=5)
    short = P.diffeqsolve(eq, P.Euler(), 0.0, 1e-3, 1e-4, y0, max_steps=5, throw=False)
    five = P.diffeqsolve(eq, P.Euler(), 0.0, 5e-4, 1e-4, y0)
    np.testing.assert_array_equal(short.ys[-1], five.ys[-1])
    # a remainder step: 0.35e-3 = 3 full steps of 1e-4 + one of 0.5e-4
    hx, hy = dom.dx
    f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
    got = P.diffeqsolve(eq, P.Euler(), 0.0, 3.5e-4, 1e-4, y0)
    assert got.stats["num_steps"] == 4
    want = O.integrate(lambda t, y, dt: O.euler_step(f, t, y, dt), y0, 0.0, 3.5e-4, 1e-4)
    np.testing.assert_allclose(got.ys[-1], want, rtol=0, atol=1e-14)


def test_nan_state_is_returned_not_raised():
    """PDEModel.solve has throw=False upstream (pde_model.py:131): divergence surfaces as NaN"""
    dom = std_domain(P, 64, 128)
    y0 = np.full((64, 128), 0.5, dtype=np.float32)
    y0[3, 3] = 1.5  # outside (0, 1): log(c / (1 - c)) is NaN there and spreads
    model = P.PDEModel(P.CahnHilliard2DPeriodic, dom, P.RK4)
    ys = model.solve(dict(kappa=0.002, mu=MU["regsol"], D=MOB["c1mc"]), y0, [0.0, 2e-6], dt0=2e-7)
    assert ys.shape == (2, 64, 128) and np.isnan(ys[-1]).any()


@pytest.mark.parametrize("case", ["ad_euler_128", "ch_rk4_fused", "ac_rk4_generic"])
def test_hipgraph_replay_equals_eager(case):
    """launch-bound sizes replay the substep loop from a captured hipGraph: same bits as eager"""
    rng = np.random.default_rng(30)
    if case == "ad_euler_128":
        dom = std_domain(P, 128, 128, h=0.02)
        eq = P.AdvectionDiffusion2D(dom, lambda t, x, y: (0.1 * np.sin(x), 0.05 * np.cos(y)), 0.1)
        y0, solver, dt, n = 0.5 + 0.01 * rng.standard_normal((128, 128)), P.Euler(), 1e-4, 101
    elif case == "ch_rk4_fused":
        dom = std_domain(P, 64, 128)
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        y0 = np.clip(0.5 + 0.05 * rng.standard_normal((2, 64, 128)), 0.05, 0.95).astype(np.float32)
        solver, dt, n = P.RK4(), 2e-7, 50
    else:
        dom = std_domain(P, 48, 40)
        eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
        y0, solver, dt, n = 0.1 * rng.standard_normal((48, 40)), P.RK4(), 1e-4, 37
    outs, names = [], []
    for mode in (-1, 1, 0):
        eng = P.HipEngine()
        eng.set_graph(mode)
        eng.set_small_persist(-1)  # this test is about the per-substep launches (the whole-step kernel has none to capture)
        outs.append(P.diffeqsolve(eq, solver, 0.0, n * dt, dt, y0, engine=eng).ys[-1])
        names.append(eng.last_kernel)
        # a second solve on the same engine reuses the cached graph
        again = P.diffeqsolve(eq, solver, 0.0, n * dt, dt, y0, engine=eng).ys[-1]
        np.testing.assert_array_equal(again, outs[-1])
        eng.close()
    assert "hipGraph" not in names[0] and "hipGraph" in names[1] or n % 16 != 0
    np.testing.assert_array_equal(outs[1], outs[0])
    np.testing.assert_array_equal(outs[2], outs[0])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(100, 100), (64, 64), (48, 40), (72, 200), (8, 16), (130, 132)])
def test_ragged_tiles_match_generic_and_oracle(engine, dtype, shape):
    """grids the tiles do not divide (the reference notebooks use 64^2, 100^2 ...) run ragged tiles"""
    rng = np.random.default_rng(31)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    hx, hy = dom.dx
    engine.set_small_persist(-1)  # the tiled kernels are the subject (small grids would take the whole-step kernel)
    try:
        _ragged_body(engine, dtype, shape, rng, dom)
    finally:
        engine.set_small_persist(0)


def _ragged_body(engine, dtype, shape, rng, dom):
    nx, ny = shape
    hx, hy = dom.dx
    for cls, fn, kind in ((P.CahnHilliard2DPeriodic, O.ch_rhs_fd, "c"), (P.AllenCahn2DPeriodic, O.ac_rhs_fd, "sym")):
        mu_fn, mob_fn = (MU["regsol"], MOB["c1mc"]) if kind == "c" else (MU["cubic"], MOB["one_plus_sq"])
        eq = cls(dom, 0.002, mu_fn, mob_fn)
        u = white_noise_state(rng, (2, nx, ny), dtype, kind)
        got = eq.rhs(u, 0.0)
        assert "tiled" in engine.last_kernel, (shape, engine.last_kernel)
        for b in range(2):
            assert rel_l2(got[b], fn(u[b], hx, hy, 0.002, mu_fn, mob_fn)) < TOL[np.dtype(dtype)]
        # RK4 through the fused stage pairs == per-stage generic kernels to rounding
        sol = P.diffeqsolve(eq, P.RK4(), 0.0, 4 * 2e-7, 2e-7, u)
        assert "pair" in sol.stats["kernel"] or "rk4_quad" in sol.stats["kernel"]
        eng = P.HipEngine()
        eng.set_kernel_path(L.PATH_GENERIC)
        ref = P.diffeqsolve(eq, P.RK4(), 0.0, 4 * 2e-7, 2e-7, u, engine=eng).ys[-1]
        eng.close()
        inc, inc_ref = sol.ys[-1].astype(np.float64) - u, ref.astype(np.float64) - u
        if dtype is np.float64:
            assert rel_l2(inc, inc_ref) < 1e-10, (shape, cls.__name__)
        else:
            # fp32: 4 substeps of 2e-7 move the state by ~1e-5, a few hundred ulps; kernels that round the RK
            # combination in a different order differ by an ulp of the STATE, so compare states
            assert np.max(np.abs(sol.ys[-1].astype(np.float64) - ref)) < 5e-7, (shape, cls.__name__)
            assert rel_l2(inc, inc_ref) < 2e-2, (shape, cls.__name__)


def test_linear_logit_class_matches_the_cubic_one():
    """CL_LOGIT1 (csrc/closures.hpp): the regular-solution closure written with 2 coefficients runs the
    shorter in-kernel form (linear polynomial part, kappa and 1/h^2 folded into the Laplacian weights: the
    same expression re-associated), with 4 coefficients (two of them zero) the literal cubic one.  Equal to a
    few ulp of the fp32 state after 8 substeps."""
    from pde_opt_amd.numerics.closures import LOGIT_PRIOR, POLY, ClosureDesc

    rng = np.random.default_rng(23)
    dom = std_domain(P, 128, 256)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((2, 128, 256)), 0.05, 0.95).astype(np.float32)
    outs = []
    for coef in ((3.0, -6.0), (3.0, -6.0, 0.0, 0.0)):
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, ClosureDesc(POLY, LOGIT_PRIOR, coef), MOB["c1mc"])
        eng = P.HipEngine()
        eng.set_small_persist(-1)  # the tiled whole-substep kernel (whose closure classes are the subject)
        sol = P.diffeqsolve(eq, P.RK4(), 0.0, 8 * 2e-7, 2e-7, y0, engine=eng)
        eng.close()
        assert "rk4_quad" in sol.stats["kernel"]
        outs.append(sol.ys[-1])
    eps = np.finfo(np.float32).eps
    np.testing.assert_allclose(outs[0], outs[1], rtol=0, atol=8 * eps)
    # the traced callable closure is recognised as the same 2-coefficient description: same kernel, same bits
    eng = P.HipEngine()
    eng.set_small_persist(-1)
    ref = P.diffeqsolve(P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"]), P.RK4(), 0.0, 8 * 2e-7, 2e-7, y0, engine=eng).ys[-1]
    eng.close()
    np.testing.assert_array_equal(outs[0], ref)


@pytest.mark.parametrize("mob", ["one_plus_sq", "const015"])  # general / constant-mobility instantiation
@pytest.mark.parametrize("shape,batch", [((512, 512), 2), ((100, 100), 3), ((64, 128), 5), ((48, 40), 1)])
def test_allen_cahn_single_pass_rk4_equals_stage_pairs(shape, batch, mob):
    """csrc/stencil_fused_ac4.hpp (all four RK4 stages in one pass, fp32) against the stage-pair kernels and
    the per-stage kernels: the same arithmetic per stage, results equal to rounding; ragged grids included."""
    rng = np.random.default_rng(29)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB[mob])
    u = white_noise_state(rng, (batch, nx, ny), np.float32, "sym")
    outs = {}
    for fuse in (0, 1, -1):  # auto (single pass), stage pairs, one launch per stage
        eng = P.HipEngine()
        eng.set_fuse_stages(fuse)
        eng.set_small_persist(-1)
        eng.configure(dtype=np.float32, batch=batch, **eq._engine_problem())
        eng.set_env_params(0, kappa=0.002 * (1.0 + 0.1 * np.arange(batch)))
        eng.set_state(u)
        eng.advance(L.INT_RK4, 5e-5, 40)
        outs[fuse] = eng.get_state()
        assert ("rk4_quad" in eng.last_kernel) == (fuse == 0), eng.last_kernel
        eng.close()
    assert np.isfinite(outs[0]).all()
    for other in (1, -1):
        assert rel_l2(outs[0].astype(np.float64) - u, outs[other].astype(np.float64) - u) < 2e-5
    hx, hy = dom.dx
    f = lambda t, v: O.ac_rhs_fd(v, hx, hy, 0.002, MU["cubic"], MOB[mob])
    ref = u[0].astype(np.float64)
    for i in range(40):
        ref = O.rk4_step(f, 0.0, ref, 5e-5)
    assert rel_l2(outs[0][0].astype(np.float64) - u[0], ref - u[0]) < 5e-4


@pytest.mark.parametrize("closures", [("regsol", "c1mc"), ("cubic", "one_plus_sq"), ("regsol4", "c1mc")])
@pytest.mark.parametrize("shape,batch", [((32, 128), 3), ((256, 384), 2), ((1024, 1024), 2), ((64, 256), 5), ((128, 192), 2), ((64, 64), 3)])
def test_cahn_hilliard_single_pass_rk4_equals_stage_pairs(shape, batch, closures):
    """csrc/stencil_fused_ch4.hpp (all four RK4 stages of Cahn-Hilliard in one pass over HBM, fp32, three LDS arrays,
    tile + 8 halo) against the stage-pair kernels: the same mu form, face fluxes, divergence and update association
    -- bitwise -- and against the oracle.  The smallest grids are one workgroup tile whose halo is the tile itself;
    (128, 192) and (64, 64) run the 64 x 64 tile."""
    mu, mob = closures
    if mu not in MU:
        pytest.skip("closure not in the test catalogue")
    rng = np.random.default_rng(31)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU[mu], MOB[mob])
    u = white_noise_state(rng, (batch, nx, ny), np.float32, "c")
    dt = 2e-7 if mob == "c1mc" else 2e-8
    kappas = 0.002 * (1.0 + 0.1 * np.arange(batch))
    outs = {}
    for fuse in (0, 1):  # auto: the whole-substep kernel; 1: stage pairs
        eng = P.HipEngine()
        eng.set_fuse_stages(fuse)
        eng.set_small_persist(-1)
        eng.set_graph(-1)
        eng.configure(dtype=np.float32, batch=batch, **eq._engine_problem())
        eng.set_env_params(0, kappa=kappas)
        eng.set_state(u)
        eng.advance(L.INT_RK4, dt, 7)
        outs[fuse] = eng.get_state()
        assert ("rk4_quad" in eng.last_kernel) == (fuse == 0), eng.last_kernel
        assert ("stage_pair" in eng.last_kernel) == (fuse == 1), eng.last_kernel
        if fuse == 0:  # 32 x 128 tiles where they divide the grid, 64 x 64 tiles otherwise
            assert ("rows32" in eng.last_kernel) == (ny % 128 == 0), eng.last_kernel
        eng.close()
    assert np.isfinite(outs[0]).all() and np.any(outs[0] != u)
    np.testing.assert_array_equal(outs[0], outs[1])
    hx, hy = dom.dx
    for b in (0, batch - 1):
        f = lambda t, v: O.ch_rhs_fd(v, hx, hy, kappas[b], MU[mu], MOB[mob])
        ref = u[b].astype(np.float64)
        for i in range(7):
            ref = O.rk4_step(f, 0.0, ref, dt)
        assert np.max(np.abs(outs[0][b] - ref)) < 5e-7
        assert rel_l2(outs[0][b].astype(np.float64) - u[b], ref - u[b]) < inc_tol_f32(ref, u[b]), b


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pair_kernel_tile_heights_agree(dtype):
    """16-row (256-thread) and 32-row (512-thread, the default where 32 divides nx) tiles of the fused CH
    pair kernel run the same per-cell arithmetic; they are different template instantiations, so hipcc's
    FMA contraction may differ in an ulp on a few cells (spelling every FMA out costs 4.5 %)."""
    rng = np.random.default_rng(41)
    dom = std_domain(P, 256, 384)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    u = white_noise_state(rng, (3, 256, 384), dtype, "c")
    outs = {}
    for rows in (16, 32, 0):
        eng = P.HipEngine()
        eng.set_tile_rows(rows)
        eng.configure(dtype=dtype, batch=3, **eq._engine_problem())
        eng.set_state(u)
        eng.advance(L.INT_RK4, 2e-7, 6)
        outs[rows] = eng.get_state()
        assert f"rows{rows or 32}" in eng.last_kernel, eng.last_kernel
        eng.close()
    eps = np.finfo(dtype).eps
    np.testing.assert_allclose(outs[32], outs[16], rtol=0, atol=4 * eps)
    assert np.mean(outs[32] != outs[16]) < 0.01
    np.testing.assert_array_equal(outs[0], outs[32])


# ------------------------------------------------------------------ the reference's convergence harness
@pytest.mark.parametrize("kind", ["ac", "ch", "ad"])
def test_rhs_convergence_through_check_convergence(kind):
    """tests/test_rhs_convergence.py:14-77 as written upstream -- equation class + symbolic class + the two argument
    dictionaries into ``check_convergence`` -- on the build's counterparts of ``numerics/utils/testing.py`` and
    ``numerics/symbolic``; the advection-diffusion row rides the same harness."""
    import sympy as sp

    from pde_opt_amd.numerics.symbolic import (SymbolicAdvectionDiffusion2D, SymbolicAllenCahn2DPeriodic,
                                               SymbolicCahnHilliard2DPeriodic)
    from pde_opt_amd.numerics.utils.testing import check_convergence, convergence_slope

    x, y, t = sp.symbols("x y t", real=True)
    u_star = sp.sin(2 * x) * sp.cos(3 * y) * sp.exp(-0.7 * t)
    mu_sym = lambda u: u**3 - u
    R_sym = lambda u: 1 + u**2
    Ns = [32, 64, 128, 256, 512]
    if kind == "ac":
        args = (P.AllenCahn2DPeriodic, SymbolicAllenCahn2DPeriodic,
                {"kappa": 1e-2, "mu": mu_sym, "R": R_sym, "derivs": "fd"},
                {"kappa": 1e-2, "mu_sym": mu_sym, "R_sym": R_sym, "u_star": u_star})
    elif kind == "ch":
        args = (P.CahnHilliard2DPeriodic, SymbolicCahnHilliard2DPeriodic,
                {"kappa": 1e-2, "mu": mu_sym, "D": R_sym, "derivs": "fd"},
                {"kappa": 1e-2, "mu_sym": mu_sym, "D_sym": R_sym, "u_star": u_star})
    else:
        vel_sym = lambda xs, ys, ts: (sp.Rational(3, 5) + sp.Rational(3, 10) * sp.sin(xs) * sp.cos(2 * ys),
                                      -sp.Rational(2, 5) + sp.Rational(1, 5) * sp.cos(3 * xs) * sp.sin(ys))
        vel = lambda tt, xx, yy: (0.6 + 0.3 * np.sin(xx) * np.cos(2 * yy), -0.4 + 0.2 * np.cos(3 * xx) * np.sin(yy))
        args = (P.AdvectionDiffusion2D, SymbolicAdvectionDiffusion2D, {"velocity_fn": vel, "D": 0.05},
                {"velocity_sym": vel_sym, "D": 0.05, "u_star": u_star})
    numeric_args, symbolic_args = dict(args[2]), dict(args[3])
    dxs, errors = check_convergence(args[0], args[1], numeric_args, symbolic_args, Ns, 2 * np.pi)
    assert numeric_args == args[2] and symbolic_args == args[3]  # the caller's dictionaries are left alone
    np.testing.assert_allclose(convergence_slope(dxs, errors), 2.0, rtol=0.1)
    assert errors[-1] < 2e-3 and all(a > b for a, b in zip(errors, errors[1:]))


def test_fp64_logit_is_libm_class_on_the_gpu():
    """The fp64 regular-solution closure log(c / (1 - c)) is hand-built (frexp of c and 1 - c, one Newton-refined
    division, nine-term atanh series: csrc/closures.hpp).  Through the Allen-Cahn kernel with kappa = 0 and R = 1 the
    right-hand side is -mu_h(u): compared with numpy's correctly rounded long-double log over (1e-9, 1 - 1e-9),
    dense near 1/2 (where the logit vanishes and relative error is hardest) and near the ends."""
    rng = np.random.default_rng(99)
    n = 256
    c = np.concatenate([rng.uniform(1e-9, 1 - 1e-9, n * n // 2), 0.5 + 1e-2 * rng.standard_normal(n * n // 4),
                        10.0 ** rng.uniform(-9, -1, n * n // 8), 1.0 - 10.0 ** rng.uniform(-9, -1, n * n // 8)])
    c = np.clip(c, 1e-9, 1 - 1e-9).reshape(n, n)
    dom = std_domain(P, n, n)
    eq = P.AllenCahn2DPeriodic(dom, 0.0, lambda u: np.log(u / (1 - u)), lambda u: np.ones_like(u))
    got = -eq.rhs(c, 0.0)
    cl = c.astype(np.longdouble)
    want = np.log(cl / (1 - cl))  # 64-bit mantissa: the reference for a 53-bit result
    err = np.abs(got.astype(np.longdouble) - want)
    ulp = np.spacing(np.abs(want.astype(np.float64)))
    assert float(np.max(err / np.maximum(ulp, 2.3e-16))) < 4.0, float(np.max(err / np.maximum(ulp, 2.3e-16)))
    assert float(err.max()) < 4e-15
