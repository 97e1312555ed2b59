"""PDEEnv / VectorPDEEnv protocol on the GPU (the reference has zero PDEEnv coverage; the
protocol is taken from pde_opt/pde_env.py:217-317)."""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from util import MOB, MU, rel_l2, std_domain

pytestmark = pytest.mark.gpu


def _reset(domain, seed=0):
    rng = np.random.default_rng(seed)
    return np.clip(0.5 + 0.01 * rng.standard_normal(domain.points), 0.05, 0.95)


def _env_kwargs(dom, solver_type=None, solver_parameters=None, step_dt=2e-6, numeric_dt=2e-7):
    return dict(
        equation_type=P.CahnHilliard2DPeriodic,
        domain=dom,
        solver_type=solver_type or P.RK4,
        end_time=3 * step_dt,
        step_dt=step_dt,
        numeric_dt=numeric_dt,
        state_to_observation_func=lambda s: np.clip(s * 255, 0, 255).astype(np.uint8)[None],
        reward_function=lambda s: float(np.var(s)),
        reset_func=_reset,
        reset_control_value=0.002,
        update_control_value=lambda off, old: old + off,
        update_control_parameter=lambda old, new: new,
        action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -0.0005, 1: 0.0, 2: 0.0005}},
        static_equation_parameters={"mu": MU["regsol"], "D": MOB["c1mc"]},
        control_equation_parameter_name="kappa",
        solver_parameters=solver_parameters or {},
    )


def test_pdeenv_protocol_and_parity():
    dom = std_domain(P, 64, 128)
    env = P.PDEEnv(**_env_kwargs(dom))
    assert env.observation_space.shape == (1, 64, 128) and env.observation_space.dtype == np.uint8
    assert env.action_space.n == 3
    obs, info = env.reset(seed=3)
    assert obs.shape == (1, 64, 128) and info == {}
    assert env._time == 0.0 and env._control_value == 0.002
    y = _reset(dom, seed=3)
    hx, hy = dom.dx
    kappas = []
    terminated = False
    steps = 0
    for action in (2, 1, 0):
        obs, reward, terminated, truncated, info = env.step(action)
        steps += 1
        kappas.append(env._control_value)
        f = lambda t, u, kk=env._control_value: O.ch_rhs_fd(u, hx, hy, kk, MU["regsol"], MOB["c1mc"])
        y = O.integrate(lambda t, u, dt: O.rk4_step(f, t, u, dt), y, 0.0, 2e-6, 2e-7)
        assert truncated is False and info == {}
        assert abs(reward - np.var(y)) < 1e-12
        assert np.max(np.abs(env._state - y)) < 1e-12
    np.testing.assert_allclose(kappas, [0.0025, 0.0025, 0.002])
    assert terminated is True and abs(env._time - 6e-6) < 1e-18
    env.close()


def test_pdeenv_imex_solver_params():
    dom = std_domain(P, 64, 64)
    env = P.PDEEnv(**_env_kwargs(dom, P.SemiImplicitFourierSpectral, {"A": 0.5}, step_dt=1e-5, numeric_dt=1e-6))
    env.reset(seed=1)
    _, reward, _, _, _ = env.step(1)
    y = _reset(dom, seed=1)
    hx, hy = dom.dx
    sym = O.ch_fourier_symbol(64, 64, hx, hy, 0.002)
    rhs = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, MU["regsol"], MOB["c1mc"])
    y = O.integrate(lambda t, u, dt: O.imex_step(rhs, t, u, dt, 0.5, sym), y, 0.0, 1e-5, 1e-6)
    assert np.max(np.abs(env._state - y)) < 1e-12
    with pytest.raises(ValueError, match="missing required"):
        P.PDEEnv(**{**_env_kwargs(dom), "solver_type": P.StrangSplitting})
    env.close()


def test_vector_env_matches_single_envs():
    dom = std_domain(P, 64, 128)
    kw = _env_kwargs(dom)
    venv = P.VectorPDEEnv(3, **kw, device_reward="var")
    obs, _ = venv.reset(seed=10)
    assert obs.shape == (3, 1, 64, 128)
    singles = []
    for b in range(3):
        e = P.PDEEnv(**kw)
        e.reset(seed=10 + b)
        singles.append(e)
    for actions in ([0, 1, 2], [2, 2, 0]):
        obs, rewards, term, trunc, _ = venv.step(actions)
        states = venv.states
        for b, e in enumerate(singles):
            o, r, t, _, _ = e.step(actions[b])
            np.testing.assert_array_equal(states[b], e._state)  # bitwise: batching changes nothing
            assert abs(rewards[b] - r) < 1e-15
            assert bool(term[b]) == t
    venv.close()
    for e in singles:
        e.close()


def test_vector_env_imex_with_per_environment_kappa():
    """IMEX + kappa as the per-environment control: every environment gets its own implicit operator"""
    dom = std_domain(P, 64, 64)
    kw = _env_kwargs(dom, P.SemiImplicitFourierSpectral, {"A": 0.5}, step_dt=1e-5, numeric_dt=1e-6)
    venv = P.VectorPDEEnv(3, **kw, device_reward="var")
    venv.reset(seed=20)
    singles = []
    for b in range(3):
        e = P.PDEEnv(**kw)
        e.reset(seed=20 + b)
        singles.append(e)
    for actions in ([0, 1, 2], [2, 2, 0]):
        venv.step(actions)
        states = venv.states
        for b, e in enumerate(singles):
            e.step(actions[b])
            assert rel_l2(states[b] - _reset(dom, 20 + b), e._state - _reset(dom, 20 + b)) < 1e-9
    venv.step([1, 1, 1])  # equal kappas again: paired transforms
    venv.close()
    for e in singles:
        e.close()


def test_device_observation_uint8():
    dom = std_domain(P, 64, 128)
    kw = _env_kwargs(dom)
    venv = P.VectorPDEEnv(2, **kw, device_reward="mean", device_observation=(0.0, 1.0))
    venv.reset(seed=5)
    obs, rewards, _, _, _ = venv.step([1, 1])
    states = venv.states
    assert obs.shape == (2, 1, 64, 128) and obs.dtype == np.uint8
    want = np.rint(np.clip(states, 0.0, 1.0) * 255).astype(np.uint8)
    # rint ties may differ by the fp32 scaling; allow one level on a vanishing fraction of cells
    diff = np.abs(obs[:, 0].astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    np.testing.assert_allclose(rewards, states.astype(np.float64).mean(axis=(1, 2)), rtol=1e-12)
    venv.close()
    # one reused page-locked frame buffer: same frames, same array object every step
    venv = P.VectorPDEEnv(2, **kw, device_reward="mean", device_observation=(0.0, 1.0), reuse_observation_buffer=True)
    venv.reset(seed=5)
    obs1, _, _, _, _ = venv.step([1, 1])
    np.testing.assert_array_equal(obs1, obs)
    first = obs1.copy()
    obs2, _, _, _, _ = venv.step([1, 1])
    assert obs2.base is obs1.base and np.any(obs2 != first)
    eng = venv._engine
    st = eng.get_state(out=eng.pinned_empty((2, 64, 128), eng.dtype))
    np.testing.assert_array_equal(st, venv.states)
    venv.close()


def test_observations_and_state_stay_on_the_gpu_as_torch_tensors():
    """zero-copy hand-over to a consumer on the same GPU: the uint8 frames and the state field as torch CUDA
    tensors over library-owned memory (``__cuda_array_interface__``), equal to the host path bit for bit"""
    torch = pytest.importorskip("torch")
    dom = std_domain(P, 64, 128)
    kw = _env_kwargs(dom)
    host = P.VectorPDEEnv(3, **kw, device_reward="mean", device_observation=(0.0, 1.0))
    dev = P.VectorPDEEnv(3, **kw, device_reward="mean", device_observation=(0.0, 1.0), observations_on_device=True)
    host.reset(seed=5)
    dev.reset(seed=5)
    for _ in range(2):
        obs_h, rew_h, _, _, _ = host.step([1, 0, 2])
        obs_d, rew_d, _, _, _ = dev.step([1, 0, 2])
        assert isinstance(obs_d, torch.Tensor) and obs_d.is_cuda and obs_d.dtype == torch.uint8
        assert tuple(obs_d.shape) == (3, 1, 64, 128)
        np.testing.assert_array_equal(obs_d.cpu().numpy(), obs_h)
        np.testing.assert_array_equal(rew_d, rew_h)
    # a torch op consumes the frames in place on the device
    assert float(obs_d.float().mean()) == pytest.approx(float(obs_h.astype(np.float64).mean()), rel=1e-6)
    eng = dev._engine
    st = eng.state_device_array()
    assert st.shape == (3, 64, 128) and st.__cuda_array_interface__["typestr"] == np.dtype(eng.dtype).str
    t = st.torch()
    np.testing.assert_array_equal(t.cpu().numpy(), dev.states)
    t.mul_(0.5)  # the tensor aliases the library's state: the next get_state sees the change
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dev.states, 0.5 * host.states)
    with pytest.raises(ValueError):
        P.VectorPDEEnv(3, **kw, device_reward="mean", observations_on_device=True)
    host.close()
    dev.close()


def test_vector_env_over_several_engines():
    """VectorPDEEnv(engines=[...]) -- the multi-GPU product path (one engine + host thread per device, environments
    split by shard_envs, no collective) with TWO engines on device 0 and an uneven 5 = 3 + 2 split: states, rewards
    and uint8 frames equal the single-engine environment bit for bit; with observations_on_device every engine hands
    back its own tensor.  The CPU twin with 2 and 3 oracle-backed engines is tests/test_multi_device_env.py."""
    dom = std_domain(P, 64, 128)
    kw = _env_kwargs(dom)
    extra = dict(device_reward="var", device_observation=(0.0, 1.0))
    one = P.VectorPDEEnv(5, **kw, **extra)
    many = P.VectorPDEEnv(5, **kw, engines=[P.HipEngine(0), P.HipEngine(0)], **extra)
    assert many.num_devices == 2 and many.shard_bounds == [(0, 3), (3, 5)]
    one.reset(seed=40)
    many.reset(seed=40)
    for actions in ([0, 1, 2, 2, 0], [2, 2, 0, 1, 1], [1, 1, 1, 1, 1]):
        o1, r1, t1, _, _ = one.step(actions)
        o2, r2, t2, _, _ = many.step(actions)
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_array_equal(t1, t2)
        np.testing.assert_array_equal(one.states, many.states)
    assert [sh.engine.batch for sh in many._shards] == [3, 2]
    assert any(k in many._shards[1].engine.last_kernel for k in ("rk4_quad", "stage_pair", "rk4_coop")), many._shards[1].engine.last_kernel
    one.close()
    many.close()
    # host reward / observation functions and the IMEX solver (per-environment kappa: sigma relative to each
    # engine's first environment) through the same sharding
    kw = _env_kwargs(dom, P.SemiImplicitFourierSpectral, {"A": 0.5}, step_dt=1e-5, numeric_dt=1e-6)
    one = P.VectorPDEEnv(5, **kw)
    many = P.VectorPDEEnv(5, **kw, engines=[P.HipEngine(0), P.HipEngine(0), P.HipEngine(0)])
    one.reset(seed=41)
    many.reset(seed=41)
    y0 = one.states
    for actions in ([0, 1, 2, 2, 0], [2, 2, 0, 1, 1]):
        o1, r1, _, _, _ = one.step(actions)
        o2, r2, _, _, _ = many.step(actions)
        # paired transforms group environments differently per engine: equal to rounding, not bitwise
        assert rel_l2(many.states - y0, one.states - y0) < 1e-9
        np.testing.assert_allclose(r2, r1, rtol=1e-9)
    one.close()
    many.close()
    torch = pytest.importorskip("torch")
    kw = _env_kwargs(dom)
    dev = P.VectorPDEEnv(4, **kw, engines=[P.HipEngine(0), P.HipEngine(0)], observations_on_device=True, **extra)
    host = P.VectorPDEEnv(4, **kw, **extra)
    dev.reset(seed=3)
    host.reset(seed=3)
    obs_d, rew_d, *_ = dev.step([0, 1, 2, 1])
    obs_h, rew_h, *_ = host.step([0, 1, 2, 1])
    assert isinstance(obs_d, list) and len(obs_d) == 2 and all(o.is_cuda and tuple(o.shape) == (2, 1, 64, 128) for o in obs_d)
    np.testing.assert_array_equal(torch.cat(obs_d).cpu().numpy(), obs_h)
    np.testing.assert_array_equal(rew_d, rew_h)
    dev.close()
    host.close()


def test_detect_vortices_against_reference_goldens(golden):
    """rl_utils.detect_vortices on the GPU: winding map, positions, charges and counts equal the
    reference's (pde_opt/rl_utils.py:19-84) -- integers, so exactly."""
    from pde_opt_amd import rl_utils

    z = golden("vortices.npz")
    for tag in sorted({k.split("/")[0] for k in z.files}):
        psi = z[f"{tag}/psi"]
        for amp, tol in ((0.0, 0.5), (0.02, 0.5), (0.0, 1.5)):
            key = f"{tag}/amp{amp}_tol{tol}"
            r = rl_utils.detect_vortices(psi, amp_thresh=amp, tol=tol)
            np.testing.assert_array_equal(r["winding"], z[key + "/winding"])
            np.testing.assert_array_equal(r["positions"], z[key + "/positions"])
            np.testing.assert_array_equal(r["charges"], z[key + "/charges"])
            assert [r["num_vortices"], r["total_topological_charge"], r["abs_charge_count"]] == list(z[key + "/counts"])
    np.testing.assert_allclose(rl_utils.density(psi), np.abs(psi) ** 2)


def test_detect_vortices_batched_counts_only():
    """counts for a batch of resident GPE states without moving the fields (24 bytes per environment)"""
    from oracle import np_oracle as O

    n, batch = 64, 5
    rng = np.random.default_rng(3)
    xs = (np.arange(n) + 0.5) - n / 2
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    states, want = [], []
    for b in range(batch):
        psi = np.exp(-(X**2 + Y**2) / (2 * 18.0**2)).astype(complex)
        for _ in range(b):  # b vortices at random places
            cx, cy = rng.uniform(-12, 12, size=2)
            z = (X - cx) + 1j * (Y - cy)
            psi = psi * z / np.sqrt(np.abs(z) ** 2 + 1.0)
        psi = (psi + 1e-3 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))).astype(np.complex64)
        states.append(np.stack([psi.real, psi.imag], axis=-1))
        want.append(O.detect_vortices(psi, 0.01, 0.5)[1:])
    eng = P.HipEngine()
    from pde_opt_amd import _lib as L

    eng.configure(equation=L.EQ_GPE, dtype=np.float32, nx=n, ny=n, batch=batch, hx=1.0, hy=1.0)
    eng.set_state(np.stack(states))
    counts, winding = eng.detect_vortices(0.01, 0.5, want_winding=False)
    assert winding is None
    np.testing.assert_array_equal(counts, np.array(want))
    assert counts[:, 0].max() >= 3
    counts2, w2 = eng.detect_vortices(0.01, 0.5, env_first=2, env_count=2)
    np.testing.assert_array_equal(counts2, counts[2:4])
    assert w2.shape == (2, n, n) and (np.count_nonzero(w2, axis=(1, 2)) == counts2[:, 0]).all()
    eng.close()


def test_point_probes():
    """pdeopt_probe: the state at listed grid cells for every environment"""
    from pde_opt_amd import _lib as L

    rng = np.random.default_rng(8)
    cells = [(0, 0), (63, 127), (5, 17), (40, 3)]
    for dtype in (np.float32, np.float64):
        dom = std_domain(P, 64, 128)
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        y = np.clip(0.5 + 0.1 * rng.standard_normal((5, 64, 128)), 0.05, 0.95).astype(dtype)
        eng = P.HipEngine()
        eng.configure(dtype=dtype, batch=5, **eq._engine_problem())
        eng.set_state(y)
        got = eng.probe(cells)
        assert got.shape == (5, 4) and got.dtype == np.float64
        np.testing.assert_array_equal(got, np.stack([y[:, i, j] for i, j in cells], axis=1).astype(np.float64))
        np.testing.assert_array_equal(eng.probe(cells[1:3], env_first=2, env_count=2), got[2:4, 1:3])
        with pytest.raises(ValueError, match="outside"):
            eng.probe([(64, 0)])
        eng.close()
    # GPE: (re, im) per probe; 3-D: (i, j, k) triples
    eng = P.HipEngine()
    eng.configure(equation=L.EQ_GPE, dtype=np.float32, nx=32, ny=16, batch=2, hx=1.0, hy=1.0)
    psi = rng.standard_normal((2, 32, 16, 2)).astype(np.float32)
    eng.set_state(psi)
    np.testing.assert_array_equal(eng.probe([(3, 4), (31, 15)]), psi[:, [3, 31], [4, 15]].astype(np.float64))
    eq3 = P.CahnHilliard3DPeriodic(P.Domain((8, 6, 10), ((0, 1), (0, 1), (0, 1)), "d"), 0.002, MU["regsol"], MOB["c1mc"])
    u3 = rng.uniform(0.1, 0.9, size=(2, 8, 6, 10))
    eng.configure(dtype=np.float64, batch=2, **eq3._engine_problem())
    eng.set_state(u3)
    np.testing.assert_array_equal(eng.probe([(7, 5, 9), (0, 2, 3)]), np.stack([u3[:, 7, 5, 9], u3[:, 0, 2, 3]], axis=1))
    eng.close()
    # as the observation of a vector environment
    dom = std_domain(P, 64, 128)
    venv = P.VectorPDEEnv(2, **_env_kwargs(dom), device_reward="var", device_observation=("probes", cells))
    venv.reset(seed=1)
    obs, rewards, _, _, _ = venv.step([1, 1])
    st = venv.states
    np.testing.assert_array_equal(obs, np.stack([st[:, i, j] for i, j in cells], axis=1).astype(np.float64))
    venv.close()
