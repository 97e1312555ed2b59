"""Time-dependent and per-environment controls on the GPU.

The reference evaluates ``lights(t0, X, Y)`` in every Strang step (numerics/solvers.py:109 ->
gross_pitaevskii.py:61,67-75): pinned here by trajectories its own ``solver.step`` produced with a moving
Gaussian spot (tests/golden/trajectories_r2.npz, oracle/gen_golden.py round2).  A batch carries one control
value per environment (k through the per-environment parameter table, e / lights / trap_factor through
per-environment potentials)."""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from test_host_controls import _env_kwargs, _spot_between
from util import MOVING_SPOT, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,kernel", [(48, "strang_rocfft_c2c"), (64, "strang_fused_lds_fft")])
@pytest.mark.parametrize("name,tscale", [("real", 1.0), ("imag", -1j)])
def test_strang_time_dependent_lights_vs_reference_golden(golden, dtype, n, kernel, name, tscale):
    z = golden("trajectories_r2.npz")
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    eq = P.GPE2DTSControl(dom, 800.0, -0.15, MOVING_SPOT, trap_factor=0.9, kinetic=True)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, tscale)
    y0 = z[f"strang_tdep/{n}/y0"].astype(dtype)
    eng = P.HipEngine()
    ts = [1e-3 * i for i in range(7)]
    sol = P.diffeqsolve(eq, solver, 0.0, 6e-3, 1e-3, y0, saveat=P.SaveAt(ts=ts), engine=eng)
    assert eng.last_kernel == kernel, eng.last_kernel
    want = z[f"strang_tdep/{n}/{name}/ys"]
    tol = 1e-11 if dtype is np.float64 else 3e-5
    for i in range(6):
        assert rel_l2(sol.ys[i + 1], want[i]) < tol, (i, rel_l2(sol.ys[i + 1], want[i]))
    # frozen at t0 (what round 1 did silently) is measurably different
    eq_f = P.GPE2DTSControl(dom, 800.0, -0.15, MOVING_SPOT, trap_factor=0.9, kinetic=True, time_dependent=False)
    frozen = P.diffeqsolve(eq_f, solver, 0.0, 6e-3, 1e-3, y0, engine=eng).ys[-1]
    assert rel_l2(frozen, want[-1]) > 100 * tol
    # one batched solve == per-field solves with the same call pattern (the source serves the whole batch)
    yb = np.stack([y0, y0[::-1].copy()])
    both = P.diffeqsolve(eq, solver, 0.0, 6e-3, 1e-3, yb, engine=eng).ys[-1]
    for b in range(2):
        np.testing.assert_array_equal(both[b], P.diffeqsolve(eq, solver, 0.0, 6e-3, 1e-3, yb[b], engine=eng).ys[-1])
    assert rel_l2(both[0], want[-1]) < tol
    eng.close()


def test_pdeenv_gpe_lights_control_matches_oracle():
    """the RL stirring control: update_control_parameter returns a callable of local time (pde_env.py:281-287)"""
    n, step_dt, dt = 64, 3e-3, 1e-3
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    kw = _env_kwargs(dom, "lights", dict(k=800.0, e=0.1, trap_factor=1.0, kinetic=True), 0.0, _spot_between(step_dt))
    env = P.PDEEnv(**kw)
    env.reset(seed=2)
    X, Y = dom.mesh()
    y = kw["reset_func"](dom, seed=2)
    h = dom.dx[0]
    ikx, iky = O.fft_wavenumbers(n, n, h, h)
    a_term = 0.5j * (ikx**2 + iky**2)
    pos = 0.0
    for action in (2, 2, 0):
        env.step(action)
        new = pos + {0: -1.0, 1: 0.0, 2: 1.0}[action]
        lights = _spot_between(step_dt)(pos, new)
        b = lambda t, yy: O.gpe_b_terms(yy, X, Y, 800.0, 0.1, 1.0, lights(t, X, Y))
        for i in range(3):
            y = O.strang_step(b, i * dt, y, dt, a_term, h, 1.0)
        pos = new
        assert rel_l2(env._state, y) < 1e-11, rel_l2(env._state, y)
    assert env._engine.last_kernel == "strang_fused_lds_fft"
    env.close()


@pytest.mark.parametrize("control", ["k", "lights", "e"])
def test_vector_env_gpe_matches_single_envs(control):
    """VectorPDEEnv vs one PDEEnv per environment with GPE controls: bitwise (batching changes nothing)"""
    n, step_dt = 64, 3e-3
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    static = dict(k=800.0, e=0.1, lights=lambda t, x, y: 0.02 * x * y, trap_factor=1.0, kinetic=True)
    static.pop(control)
    if control == "k":
        reset_value, update, mapping = 800.0, (lambda old, new: new), {0: -100.0, 1: 0.0, 2: 150.0}
    elif control == "e":
        reset_value, update, mapping = 0.1, (lambda old, new: new), {0: -0.05, 1: 0.0, 2: 0.2}
    else:
        reset_value, update, mapping = 0.0, _spot_between(step_dt), {0: -1.0, 1: 0.0, 2: 1.0}
    kw = _env_kwargs(dom, control, static, reset_value, update)
    kw["action_space_config"] = {"type": "discrete", "num_actions": 3, "action_mapping": mapping}
    venv = P.VectorPDEEnv(3, **kw)
    venv.reset(seed=5)
    singles = []
    for b in range(3):
        e = P.PDEEnv(**kw)
        e.reset(seed=5 + b)
        singles.append(e)
    for actions in ([0, 1, 2], [2, 0, 0]):
        venv.step(actions)
        states = venv.states
        for b, e in enumerate(singles):
            e.step(actions[b])
            np.testing.assert_array_equal(states[b], e._state)
        assert np.abs(states[0] - states[2]).max() > 1e-6
    assert venv._engine.last_kernel == "strang_fused_lds_fft"
    venv.close()
    for e in singles:
        e.close()


def test_config4_size_per_environment_k_vs_oracle():
    """512^2 complex64 with a different interaction strength per environment (the config-4 RL case)"""
    n, batch, nsub, dt = 512, 6, 3, 1e-3
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    eq = P.GPE2DTSControl(dom, 1000.0, 0.0, lambda t, x, y: 0.0, trap_factor=1.0, kinetic=True)
    X, Y = dom.mesh()
    psi = np.exp(-(X**2 + Y**2) / (2 * 4.0**2)).astype(complex)
    psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * dom.dx[0] ** 2)
    y0 = np.repeat(np.stack([psi.real, psi.imag], axis=-1)[None].astype(np.float32), batch, axis=0)
    ks = 600.0 + 150.0 * np.arange(batch)
    eng = P.HipEngine()
    eng.configure(dtype=np.float32, batch=batch, **eq._engine_problem())
    eqs = [P.GPE2DTSControl(dom, float(k), 0.0, eq.lights, trap_factor=1.0, kinetic=True) for k in ks]
    P.GPE2DTSControl._engine_upload_batch(eng, eqs, 0.0, nsub * dt)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    solver.configure_engine(eng, eq)
    eng.set_state(y0)
    eng.advance(L.INT_STRANG, dt, nsub)
    out = eng.get_state()
    eng.close()
    for b in (0, batch - 1):
        bt = lambda t, yy, k=ks[b]: O.gpe_b_terms(yy, X, Y, k, 0.0, 1.0, 0.0)
        ref = y0[b].astype(np.float64)
        for i in range(nsub):
            ref = O.strang_step(bt, i * dt, ref, dt, eq.A_term, eq.dx, 1.0)
        assert rel_l2(out[b], ref) < 2e-5, (b, rel_l2(out[b], ref))
    assert rel_l2(out[0], out[-1]) > 1e-4


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,kernel", [(48, "strang_rocfft_c2c"), (128, "strang_fused_lds_fft")])
def test_gaussian_spots_in_kernel_vs_oracle(dtype, n, kernel):
    """GaussianSpots lights evaluated by the Strang kernels (pdeopt_set_gpe_spots) == the reference expression
    lights(t0, X, Y) at every substep's t0 == the same control sampled on the host per substep."""
    dt, nsub = 1e-3, 5
    spots = P.GaussianSpots.moving(25.0, (-2.0, 0.5), (1.5, -1.0), nsub * dt, 1.3) + P.GaussianSpots.single((10.0, 2000.0), 3.0, (0.0, -300.0), 0.9)
    dom = P.Domain((n, n), ((-12.0, 12.0), (-10.0, 10.0)), "dimensionless")
    X, Y = dom.mesh()
    rng = np.random.default_rng(3)
    states = []
    for b in range(3):
        psi = np.exp(-(X**2 + Y**2) / (16.0 + b)) * np.exp(0.05j * rng.standard_normal((n, n)))
        psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * dom.dx[0] ** 2)
        states.append(np.stack([psi.real, psi.imag], axis=-1))
    y0 = np.stack(states).astype(dtype)
    outs = {}
    for name, lights in (("kernel", spots), ("host", lambda t, x, y: spots(t, x, y))):
        eq = P.GPE2DTSControl(dom, 800.0, -0.15, lights, trap_factor=0.9, kinetic=True)
        solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
        eng = P.HipEngine()
        outs[name] = P.diffeqsolve(eq, solver, 0.0, nsub * dt, dt, y0, engine=eng).ys[-1]
        assert eng.last_kernel == kernel, eng.last_kernel
        eng.close()
    tol = 1e-11 if dtype is np.float64 else 3e-5
    b_t = lambda t, yy: O.gpe_b_terms(yy, X, Y, 800.0, -0.15, 0.9, spots(t, X, Y))
    for b in range(3):
        want = y0[b].astype(np.float64)
        for i in range(nsub):
            want = O.strang_step(b_t, i * dt, want, dt, eq.A_term, eq.dx, 1.0)
        assert rel_l2(outs["kernel"][b], want) < tol, rel_l2(outs["kernel"][b], want)
        assert rel_l2(outs["host"][b], want) < tol
    assert rel_l2(outs["kernel"], outs["host"]) < tol


def test_vector_env_spot_controls_match_single_envs():
    """every environment steers its own spot; batched == one PDEEnv per environment, bitwise"""
    n, step_dt = 64, 3e-3
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    update = lambda old, new: P.GaussianSpots.moving(25.0, (old, 0.0), (new, 0.0), step_dt, 1.2)
    kw = _env_kwargs(dom, "lights", dict(k=800.0, e=0.1, trap_factor=1.0, kinetic=True), 0.0, update)
    venv = P.VectorPDEEnv(3, **kw)
    venv.reset(seed=5)
    singles = []
    for b in range(3):
        e = P.PDEEnv(**kw)
        e.reset(seed=5 + b)
        singles.append(e)
    for actions in ([0, 1, 2], [2, 0, 0]):
        venv.step(actions)
        states = venv.states
        for b, e in enumerate(singles):
            e.step(actions[b])
            np.testing.assert_array_equal(states[b], e._state)
        assert np.abs(states[0] - states[2]).max() > 1e-6
    venv.close()
    for e in singles:
        e.close()
    # the vortex census as a device reward equals rl_utils.detect_vortices on the fetched states
    from pde_opt_amd.rl_utils import detect_vortices

    venv = P.VectorPDEEnv(3, **kw, fetch_observations=False, device_reward=("vortices", 1e-4, 0.5))
    venv.reset(seed=5)
    for _ in range(3):
        obs, rewards, _, _, _ = venv.step([2, 2, 0])
    assert obs is None
    st = venv.states
    want = [detect_vortices(st[b, ..., 0] + 1j * st[b, ..., 1], amp_thresh=1e-4)["num_vortices"] for b in range(3)]
    np.testing.assert_array_equal(rewards, want)
    venv.close()


@pytest.mark.parametrize("solver", ["euler", "rk4", "tsit5"])
def test_advection_time_dependent_velocity_vs_oracle(solver):
    """velocity_fn(t, x, y) re-sampled at every Runge-Kutta stage time"""
    n = 64
    dom = P.Domain((n, n), ((0.0, 1.0), (0.0, 1.0)), "dimensionless")
    vel = lambda t, x, y: ((1.0 + 20.0 * t) * np.sin(2 * np.pi * y), -0.5 * np.cos(2 * np.pi * x) * (1.0 - 10.0 * t))
    eq = P.AdvectionDiffusion2D(dom, vel, 0.01)
    y0 = 0.5 + 0.1 * np.random.default_rng(0).standard_normal((2, n, n))
    dt, nsub = 1e-3, 5
    s = {"euler": P.Euler(), "rk4": P.RK4(), "tsit5": P.Tsit5()}[solver]
    got = P.diffeqsolve(eq, s, 0.0, nsub * dt, dt, y0).ys[-1]
    hx, hy = dom.dx
    f = lambda t, u: O.ad_rhs_fd(u, hx, hy, *eq.face_velocities(t), 0.01)
    for b in range(2):
        want = y0[b]
        for i in range(nsub):
            if solver == "euler":
                want = O.euler_step(f, i * dt, want, dt)
            elif solver == "rk4":
                want = O.rk4_step(f, i * dt, want, dt)
            else:
                want = O.tsit5_step(f, i * dt, want, dt)[0]
        assert rel_l2(got[b] - y0[b], want - y0[b]) < 1e-11, rel_l2(got[b] - y0[b], want - y0[b])
    # frozen at t0 is different
    eq_f = P.AdvectionDiffusion2D(dom, vel, 0.01, time_dependent=False)
    frozen = P.diffeqsolve(eq_f, s, 0.0, nsub * dt, dt, y0).ys[-1]
    assert rel_l2(frozen - y0, got - y0) > 1e-3


def test_source_exceptions_surface_as_python_exceptions():
    n = 64
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")

    def lights(t, x, y):
        if t > 1.5e-3:
            raise RuntimeError("control blew up")
        return 0.1 * t * x

    eq = P.GPE2DTSControl(dom, 800.0, 0.0, lights, time_dependent=True, kinetic=True)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    y0 = np.zeros((n, n, 2))
    y0[..., 0] = 1.0 / 24.0
    with pytest.raises(RuntimeError, match="control blew up"):
        P.diffeqsolve(eq, solver, 0.0, 4e-3, 1e-3, y0, engine=P.HipEngine())


def test_advection_diffusion_manufactured_solution_slope():
    """SURVEY section 8 a15's pin: sympy manufactured solution, second-order slope (2.0 +- 10 %), in the
    style of the reference's tests/test_rhs_convergence.py:14-77, on the HIP kernel."""
    import sympy as sp
    from sympy.utilities.lambdify import lambdify

    x, y, t = sp.symbols("x y t", real=True)
    u = sp.sin(2 * x) * sp.cos(3 * y) * sp.exp(-0.7 * t)
    vx = sp.Rational(3, 5) + sp.Rational(3, 10) * sp.sin(x) * sp.cos(2 * y) * (1 + t)
    vy = -sp.Rational(2, 5) + sp.Rational(1, 5) * sp.cos(3 * x) * sp.sin(y)
    Dc = 0.05
    exact = -(sp.diff(vx * u, x) + sp.diff(vy * u, y)) + Dc * (sp.diff(u, x, 2) + sp.diff(u, y, 2))
    u_fn, ex_fn = lambdify((x, y, t), u, "numpy"), lambdify((x, y, t), exact, "numpy")
    vx_fn, vy_fn = lambdify((x, y, t), vx, "numpy"), lambdify((x, y, t), vy, "numpy")
    vel = lambda tt, xx, yy: (vx_fn(xx, yy, tt), vy_fn(xx, yy, tt) + 0 * xx)
    for t_eval in (0.0, 0.3):
        hs, errs = [], []
        for n in (32, 64, 128, 256, 512):
            Lb = 2 * np.pi
            dom = P.Domain((n, n), ((-Lb / 2, Lb / 2), (-Lb / 2, Lb / 2)), "dimensionless")
            X, Y = dom.mesh()
            eq = P.AdvectionDiffusion2D(dom, vel, Dc)
            got = eq.rhs(u_fn(X, Y, t_eval), t_eval)
            ex = ex_fn(X, Y, t_eval)
            errs.append(np.sqrt(np.sum((got - ex) ** 2)) / np.sqrt(np.sum(ex**2)))
            hs.append(dom.dx[0])
        slope = np.polyfit(np.log(hs), np.log(errs), 1)[0]
        np.testing.assert_allclose(slope, 2.0, rtol=0.1)
        assert errs[-1] < 1e-3
