"""Time-dependent and per-environment controls: host logic on the oracle-backed engine double, and the
oracle itself against goldens the reference's own ``solver.step`` produced with a time-dependent
``lights`` (oracle/gen_golden.py round2).

The reference evaluates ``lights(t0, X, Y)`` in EVERY Strang step (numerics/solvers.py:109 ->
gross_pitaevskii.py:61,67-75); its PDEEnv hands ``update_control_parameter(old, new)`` -- possibly a
callable of time -- to the equation every environment step (pde_env.py:281-287)."""
import numpy as np
import pytest

import pde_opt_amd as P
from fake_engine import OracleEngine
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import MOVING_SPOT, std_domain


def _gpe_mesh(n):
    h = 24.0 / n
    ax = np.linspace(-12 + h / 2, 12 - h / 2, n)
    return h, np.meshgrid(ax, ax, indexing="ij")


@pytest.mark.parametrize("n", [48, 64])
@pytest.mark.parametrize("name,tscale", [("real", 1.0), ("imag", -1j)])
def test_oracle_strang_time_dependent_lights_matches_reference(golden, n, name, tscale):
    z = golden("trajectories_r2.npz")
    h, (X, Y) = _gpe_mesh(n)
    b = lambda t, yy: O.gpe_b_terms(yy, X, Y, 800.0, -0.15, 0.9, MOVING_SPOT(t, X, Y))
    y0 = z[f"strang_tdep/{n}/y0"]
    np.testing.assert_allclose(b(0.0, y0), z[f"strang_tdep/{n}/b_terms_t0"], rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(b(3e-3, y0), z[f"strang_tdep/{n}/b_terms_t3"], rtol=1e-14, atol=1e-14)
    assert np.abs(z[f"strang_tdep/{n}/b_terms_t3"] - z[f"strang_tdep/{n}/b_terms_t0"]).max() > 1.0  # it does depend on t
    ikx, iky = O.fft_wavenumbers(n, n, h, h)
    a_term = 0.5j * (ikx**2 + iky**2)
    y = y0
    for i in range(6):
        y = O.strang_step(b, i * 1e-3, y, 1e-3, a_term, h, tscale)
        np.testing.assert_allclose(y, z[f"strang_tdep/{n}/{name}/ys"][i], rtol=0, atol=1e-13)


def test_oracle_ch3d_fourier_matches_reference(golden):
    z = golden("ch3d_fourier.npz")
    mu = lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c)
    tags = sorted(k[: -len("/rhs")] for k in z.files if k.endswith("/rhs"))
    assert len(tags) == 3
    for tag in tags:
        nx, ny, nz = (int(v) for v in tag.split("_")[0].split("x"))
        got = O.ch3d_rhs_fourier(z[tag + "/u"], 0.01, 0.01, 0.012, 0.002, mu, lambda c: c * (1 - c))
        np.testing.assert_allclose(got, z[tag + "/rhs"], rtol=0, atol=1e-12 * np.abs(z[tag + "/rhs"]).max())


def _gpe(n, lights, **kw):
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    return dom, P.GPE2DTSControl(dom, 800.0, -0.15, lights, trap_factor=0.9, kinetic=True, **kw)


def test_driver_evaluates_lights_at_every_substep(golden):
    """diffeqsolve registers a per-substep source for a time-dependent control and reproduces the
    reference trajectory; a constant control is uploaded once; time_dependent=False freezes it."""
    z = golden("trajectories_r2.npz")
    n = 48
    dom, eq = _gpe(n, MOVING_SPOT)
    y0 = z[f"strang_tdep/{n}/y0"]
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    eng = OracleEngine()
    ts = [1e-3 * i for i in range(7)]
    sol = P.diffeqsolve(eq, solver, 0.0, 6e-3, 1e-3, y0, saveat=P.SaveAt(ts=ts), engine=eng)
    np.testing.assert_allclose(sol.ys[1:], z[f"strang_tdep/{n}/real/ys"], rtol=0, atol=1e-13)
    assert ("aux_time_fn", L.AUX_GPE_POTENTIAL, False) in eng.calls
    evals = [c[2] for c in eng.calls if c[0] == "aux_eval"]
    np.testing.assert_allclose(evals, [1e-3 * i for i in range(6)], atol=1e-15)  # t0 of each substep, once

    # constant control: no source, one upload
    dom, eq_c = _gpe(n, lambda t, x, y: 0.05 * x)
    eng = OracleEngine()
    P.diffeqsolve(eq_c, solver, 0.0, 3e-3, 1e-3, y0, engine=eng)
    assert not [c for c in eng.calls if c[0] in ("aux_time_fn", "aux_eval")]

    # explicit override: frozen at t0 although lights depends on t
    dom, eq_f = _gpe(n, MOVING_SPOT, time_dependent=False)
    eng = OracleEngine()
    frozen = P.diffeqsolve(eq_f, solver, 0.0, 6e-3, 1e-3, y0, engine=eng).ys[-1]
    assert not [c for c in eng.calls if c[0] == "aux_eval"]
    assert np.abs(frozen - z[f"strang_tdep/{n}/real/ys"][-1]).max() > 1e-6


def _env_kwargs(dom, control_name, static, reset_value, update_parameter, step_dt=3e-3, numeric_dt=1e-3):
    X, Y = dom.mesh()

    def reset(domain, seed=0):
        psi = np.exp(-(X**2 + Y**2) / (2 * (3.0 + 0.1 * seed) ** 2)) * np.exp(-0.2j * Y)
        psi = psi / np.sqrt(np.sum(np.abs(psi) ** 2) * domain.dx[0] ** 2)
        return np.stack([psi.real, psi.imag], axis=-1)

    return dict(
        equation_type=P.GPE2DTSControl, domain=dom, solver_type=P.StrangSplitting, end_time=2 * step_dt,
        step_dt=step_dt, numeric_dt=numeric_dt, state_to_observation_func=lambda s: s, reward_function=lambda s: 0.0,
        reset_func=reset, reset_control_value=reset_value, update_control_value=lambda off, old: old + off,
        update_control_parameter=update_parameter,
        action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -1.0, 1: 0.0, 2: 1.0}},
        static_equation_parameters=static, control_equation_parameter_name=control_name,
        solver_parameters={"time_scale": 1.0},
    )


def _spot_between(step_dt):
    """the RL stirring control: a spot whose x position moves linearly from the old to the new control
    value during the environment step (local time restarts at 0, pde_env.py:296-297)"""
    def update(old, new):
        return lambda t, x, y: 25.0 * np.exp(-((x - (old + (new - old) * t / step_dt)) ** 2 + y**2) / 3.0)
    return update


def test_pdeenv_time_dependent_lights_control():
    n, step_dt, dt = 32, 3e-3, 1e-3
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    kw = _env_kwargs(dom, "lights", dict(k=800.0, e=0.1, trap_factor=1.0, kinetic=True), 0.0, _spot_between(step_dt))
    eng = OracleEngine()
    env = P.PDEEnv(**kw, engine=eng)
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
            y = O.strang_step(b, i * dt, y, dt, a_term, h, 1.0)  # local time restarts at 0 every env step
        pos = new
        np.testing.assert_allclose(env._state, y, rtol=0, atol=1e-13)
    evals = [c[2] for c in eng.calls if c[0] == "aux_eval"]
    np.testing.assert_allclose(evals, [0.0, dt, 2 * dt] * 3, atol=1e-15)


@pytest.mark.parametrize("control", ["k", "lights", "e"])
def test_vector_env_gpe_controls_travel_with_their_environment(control):
    """every environment of a VectorPDEEnv integrates with ITS control (environment 0's
    potential and k were silently used for the whole batch)"""
    n, step_dt = 32, 3e-3
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    static = dict(k=800.0, e=0.1, lights=lambda t, x, y: 0.02 * x * y, trap_factor=1.0, kinetic=True)
    static.pop(control)
    if control == "k":
        reset_value, update = 800.0, (lambda old, new: new)
        mapping = {0: -100.0, 1: 0.0, 2: 150.0}
    elif control == "e":
        reset_value, update = 0.1, (lambda old, new: new)
        mapping = {0: -0.05, 1: 0.0, 2: 0.2}
    else:
        reset_value, update = 0.0, _spot_between(step_dt)
        mapping = {0: -1.0, 1: 0.0, 2: 1.0}
    kw = _env_kwargs(dom, control, static, reset_value, update)
    kw["action_space_config"] = {"type": "discrete", "num_actions": 3, "action_mapping": mapping}
    venv = P.VectorPDEEnv(3, **kw, engine=OracleEngine())
    venv.reset(seed=5)
    singles = []
    for b in range(3):
        e = P.PDEEnv(**kw, engine=OracleEngine())
        e.reset(seed=5 + b)
        singles.append(e)
    for actions in ([0, 1, 2], [2, 0, 0]):
        venv.step(actions)
        states = venv.states
        for b, e in enumerate(singles):
            e.step(actions[b])
            np.testing.assert_allclose(states[b], e._state, rtol=0, atol=1e-14)
        assert np.abs(states[0] - states[2]).max() > 1e-6
    # equal controls share one upload again
    venv.step([1, 1, 1])


def test_vector_env_rejects_controls_it_cannot_vary():
    from util import MOB, MU

    dom = std_domain(P, 16, 16)
    reset = lambda domain, seed=0: np.clip(0.5 + 0.01 * np.random.default_rng(seed).standard_normal(domain.points), 0.05, 0.95)
    common = dict(
        domain=dom, end_time=1.0, step_dt=2e-6, numeric_dt=1e-6, state_to_observation_func=lambda s: s,
        reward_function=lambda s: 0.0, reset_func=reset, update_control_value=lambda off, old: old + off,
        update_control_parameter=lambda old, new: new,
        action_space_config={"type": "discrete", "num_actions": 2, "action_mapping": {0: 0.0, 1: 0.0005}},
    )
    # IMEX + per-environment kappa: fourier_symbol would differ between environments
    venv = P.VectorPDEEnv(2, equation_type=P.CahnHilliard2DPeriodic, solver_type=P.SemiImplicitFourierSpectral,
                          reset_control_value=0.002, static_equation_parameters={"mu": MU["regsol"], "D": MOB["c1mc"]},
                          control_equation_parameter_name="kappa", solver_parameters={"A": 0.5}, engine=OracleEngine(), **common)
    venv.reset(seed=0)
    venv.step([0, 0])
    venv.step([0, 1])  # per-environment kappa under IMEX: per-environment implicit operator
    singles = []
    for b in range(2):
        e = P.PDEEnv(equation_type=P.CahnHilliard2DPeriodic, solver_type=P.SemiImplicitFourierSpectral,
                     reset_control_value=0.002, static_equation_parameters={"mu": MU["regsol"], "D": MOB["c1mc"]},
                     control_equation_parameter_name="kappa", solver_parameters={"A": 0.5}, engine=OracleEngine(), **common)
        e.reset(seed=b)
        e.step(0)
        e.step(b)
        np.testing.assert_allclose(venv.states[b], e._state, rtol=0, atol=1e-14)
    # derivs is structural: cannot differ inside one batch
    venv = P.VectorPDEEnv(2, equation_type=P.CahnHilliard2DPeriodic, solver_type=P.RK4,
                          reset_control_value=0.0, static_equation_parameters={"kappa": 0.002, "mu": MU["regsol"], "D": MOB["c1mc"]},
                          control_equation_parameter_name="derivs", solver_parameters={}, engine=OracleEngine(),
                          **{**common, "update_control_parameter": lambda old, new: "fd" if new == 0 else "fourier"})
    venv.reset(seed=0)
    with pytest.raises(ValueError, match="cannot differ between the environments"):
        venv.step([0, 1])
    with pytest.raises(ValueError, match="actions for"):
        venv.step([0])


def test_advection_velocity_sampled_at_every_stage_time():
    n = 24
    dom = P.Domain((n, n), ((0.0, 1.0), (0.0, 1.0)), "dimensionless")
    vel = lambda t, x, y: ((1.0 + 20.0 * t) * np.sin(2 * np.pi * y), -0.5 * np.cos(2 * np.pi * x) * (1.0 - 10.0 * t))
    eq = P.AdvectionDiffusion2D(dom, vel, 0.01)
    rng = np.random.default_rng(0)
    y0 = 0.5 + 0.1 * rng.standard_normal((n, n))
    eng = OracleEngine()
    dt = 1e-3
    got = P.diffeqsolve(eq, P.RK4(), 0.0, 4 * dt, dt, y0, engine=eng).ys[-1]
    hx, hy = dom.dx
    f = lambda t, u: O.ad_rhs_fd(u, hx, hy, *eq.face_velocities(t), 0.01)
    want = y0
    for i in range(4):
        want = O.rk4_step(f, i * dt, want, dt)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-14)
    times = sorted({round(c[2], 12) for c in eng.calls if c[0] == "aux_eval"})
    np.testing.assert_allclose(times, [0.5e-3 * i for i in range(9)], atol=1e-12)  # t, t + dt/2, t + dt
    # steady velocity: uploaded once, no source
    eq_s = P.AdvectionDiffusion2D(dom, lambda t, x, y: (np.sin(2 * np.pi * y), 0.0 * x), 0.01)
    eng = OracleEngine()
    P.diffeqsolve(eq_s, P.Euler(), 0.0, 3 * dt, dt, y0, engine=eng)
    assert not [c for c in eng.calls if c[0] in ("aux_time_fn", "aux_eval")]


def test_gaussian_spots_are_handed_to_the_kernels_not_sampled():
    """GaussianSpots lights: a table of numbers goes to the engine (pdeopt_set_gpe_spots), nothing is sampled
    on the host per substep, and the result is the reference expression evaluated at every substep's t0."""
    n, dt = 32, 1e-3
    spots = P.GaussianSpots.moving(25.0, (-2.0, 0.5), (1.5, -1.0), 5 * dt, 1.3) + P.GaussianSpots.single((10.0, 2000.0), 3.0, (0.0, -300.0), 0.9)
    assert spots.time_dependent
    dom, eq = _gpe(n, spots)
    X, Y = dom.mesh()
    np.testing.assert_allclose(eq.control(2e-3), 25.0 * np.exp(-((X + 2.0 - 700.0 * 2e-3) ** 2 + (Y - 0.5 + 300.0 * 2e-3) ** 2) / (2 * 1.69))
                               + 14.0 * np.exp(-((X - 3.0) ** 2 + (Y + 0.6) ** 2) / (2 * 0.81)), rtol=1e-13)
    rng = np.random.default_rng(3)
    psi = np.exp(-(X**2 + Y**2) / 18.0) * np.exp(0.1j * rng.standard_normal((n, n)))
    psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * dom.dx[0] ** 2)
    y0 = np.stack([psi.real, psi.imag], axis=-1)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    eng = OracleEngine()
    got = P.diffeqsolve(eq, solver, 0.0, 5 * dt, dt, y0, engine=eng).ys[-1]
    assert ("gpe_spots", (1, 2, 7)) in eng.calls
    assert not [c for c in eng.calls if c[0] in ("aux_time_fn", "aux_eval")]
    b = lambda t, yy: O.gpe_b_terms(yy, X, Y, 800.0, -0.15, 0.9, spots(t, X, Y))
    want = y0
    for i in range(5):
        want = O.strang_step(b, i * dt, want, dt, eq.A_term, eq.dx, 1.0)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)
    # a later equation without spots on the same engine removes them
    dom, eq_c = _gpe(n, lambda t, x, y: 0.05 * x)
    P.diffeqsolve(eq_c, solver, 0.0, 2 * dt, dt, y0, engine=eng)
    assert eng.spots is None
    with pytest.raises(ValueError, match="spots"):
        P.GaussianSpots([])


def test_vector_env_spot_controls_per_environment():
    n, step_dt = 32, 3e-3
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    update = lambda old, new: P.GaussianSpots.moving(25.0, (old, 0.0), (new, 0.0), step_dt, 1.2)
    kw = _env_kwargs(dom, "lights", dict(k=800.0, e=0.1, trap_factor=1.0, kinetic=True), 0.0, update)
    eng = OracleEngine()
    venv = P.VectorPDEEnv(3, **kw, engine=eng)
    venv.reset(seed=5)
    singles = []
    for b in range(3):
        e = P.PDEEnv(**kw, engine=OracleEngine())
        e.reset(seed=5 + b)
        singles.append(e)
    for actions in ([0, 1, 2], [2, 0, 0]):
        venv.step(actions)
        for b, e in enumerate(singles):
            e.step(actions[b])
            np.testing.assert_allclose(venv.states[b], e._state, rtol=0, atol=1e-14)
    assert ("gpe_spots", (3, 1, 7)) in eng.calls
    assert not [c for c in eng.calls if c[0] in ("aux_time_fn", "aux_eval")]
