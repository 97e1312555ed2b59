"""VectorPDEEnv over several devices (BASELINE.json north_star: "batched episodes shard naturally across the 8 GPUs
of one node"; SURVEY 8(e): independent units, no collective in the step).  Here on the CPU: one oracle-backed engine
double per "device" (tests/fake_engine.py, injected through the public ``engines=`` argument), 2 and 3 devices with
uneven splits, against the single-engine environment -- the sharding, the per-device host threads, the per-shard
parameter tables and the result gather are the product's code; only the arithmetic is the oracle's.  The GPU twin
(two HipEngines on device 0, bitwise against one) is tests/test_gpu_env.py::test_vector_env_over_several_engines."""
import threading

import numpy as np
import pytest

import pde_opt_amd as P
from fake_engine import OracleEngine
from pde_opt_amd.sharding import shard_envs
from util import MOB, MU, std_domain


def _reset(domain, seed=0):
    rng = np.random.default_rng(seed)
    return np.clip(0.5 + 0.01 * rng.standard_normal(domain.points), 0.05, 0.95)


def _kw(dom, solver_type=None, solver_parameters=None, step_dt=6e-7, numeric_dt=2e-7):
    return dict(
        equation_type=P.CahnHilliard2DPeriodic, domain=dom, solver_type=solver_type or P.RK4, end_time=2 * step_dt,
        step_dt=step_dt, numeric_dt=numeric_dt, state_to_observation_func=lambda s: np.clip(s * 255, 0, 255).astype(np.uint8)[None],
        reward_function=lambda s: float(np.var(s)), reset_func=_reset, reset_control_value=0.002,
        update_control_value=lambda off, old: old + off, update_control_parameter=lambda old, new: new,
        action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -0.0005, 1: 0.0, 2: 0.0005}},
        static_equation_parameters={"mu": MU["regsol"], "D": MOB["c1mc"]}, control_equation_parameter_name="kappa",
        solver_parameters=solver_parameters or {})


class _ThreadRecordingEngine(OracleEngine):
    def advance(self, *a, **k):
        self.thread = threading.get_ident()
        return super().advance(*a, **k)


@pytest.mark.parametrize("num_envs,ndev", [(4, 2), (5, 2), (5, 3), (7, 3), (3, 3)])
@pytest.mark.parametrize("mode", ["host", "device"])
def test_sharded_env_equals_single_engine_env(num_envs, ndev, mode):
    dom = std_domain(P, 16, 24)
    extra = {} if mode == "host" else dict(device_reward="var", device_observation=(0.0, 1.0))
    one = P.VectorPDEEnv(num_envs, **_kw(dom), engine=OracleEngine(), **extra)
    engines = [_ThreadRecordingEngine() for _ in range(ndev)]
    many = P.VectorPDEEnv(num_envs, **_kw(dom), engines=engines, **extra)
    assert many.num_devices == ndev and many.shard_bounds == [shard_envs(num_envs, ndev, r) for r in range(ndev)]
    o1, _ = one.reset(seed=7)
    o2, _ = many.reset(seed=7)
    np.testing.assert_array_equal(o1, o2)
    rng = np.random.default_rng(num_envs * 10 + ndev)
    for _ in range(2):
        actions = [int(a) for a in rng.integers(0, 3, num_envs)]  # per-environment kappa: every shard's table differs
        r1, r2 = one.step(actions), many.step(actions)
        for a, b in zip(r1[:4], r2[:4]):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))  # same arithmetic per environment: exact
        assert r2[0].shape == (num_envs, 1, 16, 24) and r2[0].dtype == np.uint8 and r2[1].shape == (num_envs,)
        np.testing.assert_array_equal(one.states, many.states)
    assert bool(r2[2].all())  # end_time reached on every environment
    # every device's share ran on its own host thread, and each engine holds exactly its block
    assert len({e.thread for e in engines}) == ndev and threading.get_ident() not in {e.thread for e in engines}
    assert [e.batch for e in engines] == [hi - lo for lo, hi in many.shard_bounds]
    many.close()
    # engines passed in through engines= are the caller's: close() leaves them open, and a closed
    # environment raises instead of serving its first shard only
    assert not any(getattr(e, "closed", False) for e in engines)
    with pytest.raises(RuntimeError, match="closed"):
        many.states
    with pytest.raises(RuntimeError, match="closed"):
        many.step([1] * num_envs)


def test_sharded_env_imex_per_environment_kappa():
    """IMEX + kappa as the control: sigma_b is relative to the FIRST environment OF EACH ENGINE (its uploaded symbol)"""
    dom = std_domain(P, 16, 16)
    kw = _kw(dom, P.SemiImplicitFourierSpectral, {"A": 0.5}, step_dt=2e-6, numeric_dt=1e-6)
    one = P.VectorPDEEnv(5, **kw, engine=OracleEngine())
    many = P.VectorPDEEnv(5, **kw, engines=[OracleEngine(), OracleEngine()])
    one.reset(seed=1)
    many.reset(seed=1)
    for actions in ([0, 1, 2, 2, 0], [2, 2, 1, 0, 0]):
        one.step(actions)
        many.step(actions)
        np.testing.assert_allclose(one.states, many.states, rtol=0, atol=1e-14)
    many.step([0, 0, 0, 0, 0])  # kappas 0.0015 0.002 0.002 0.0015 0.0005
    with pytest.raises(ValueError, match="positive"):
        many.step([1, 1, 1, 1, 0])  # environment 4 reaches kappa = 0: no implicit operator to scale


def test_sharded_env_errors_and_argument_checks():
    dom = std_domain(P, 16, 16)
    with pytest.raises(ValueError, match="engines for"):
        P.VectorPDEEnv(2, **_kw(dom), engines=[OracleEngine() for _ in range(3)])
    with pytest.raises(ValueError, match="not both"):
        P.VectorPDEEnv(2, **_kw(dom), engines=[OracleEngine()], engine=OracleEngine())
    with pytest.raises(ValueError, match="twice"):
        P.VectorPDEEnv(4, **_kw(dom), devices=[0, 0])
    with pytest.raises(ValueError, match="unknown device reward"):
        P.VectorPDEEnv(2, **_kw(dom), engine=OracleEngine(), device_reward="median")

    class Boom(OracleEngine):
        def advance(self, *a, **k):
            raise RuntimeError("device 1 failed")

    env = P.VectorPDEEnv(4, **_kw(dom), engines=[OracleEngine(), Boom()])
    env.reset(seed=0)
    with pytest.raises(RuntimeError, match="device 1 failed"):  # a shard's failure surfaces in the caller's thread
        env.step([1, 1, 1, 1])
    env.close()


def test_device_observation_with_a_host_reward():
    """device-formed observations no longer need a device reward: the reward function sees the fetched field"""
    dom = std_domain(P, 16, 16)
    env = P.VectorPDEEnv(2, **_kw(dom), engine=OracleEngine(), device_observation=(0.0, 1.0))
    env.reset(seed=3)
    obs, rew, *_ = env.step([1, 2])
    st = env.states
    np.testing.assert_array_equal(obs[:, 0], np.rint(np.clip(st, 0, 1) * 255).astype(np.uint8))
    np.testing.assert_allclose(rew, [np.var(s) for s in st], rtol=0, atol=1e-15)
    probes = P.VectorPDEEnv(2, **_kw(dom), engine=OracleEngine(), device_observation=("probes", [(0, 0), (3, 5)]), device_reward="mean")
    probes.reset(seed=3)
    obs, rew, *_ = probes.step([1, 2])
    np.testing.assert_array_equal(obs, probes.states[:, [0, 3], [0, 5]])


class _NullEngine(OracleEngine):
    """an engine whose advance costs nothing: what is left of VectorPDEEnv.step is the host side"""

    def advance(self, *a, **k):
        pass

    def reduce(self, op):  # the device reductions / frame quantisation are GPU calls in production, not host work
        return np.zeros(self.batch)

    def observe_u8(self, lo, hi, out=None, **k):
        if getattr(self, "_frames", None) is None:
            self._frames = np.zeros((self.batch,) + self.state_shape, np.uint8)
        return self._frames


def test_step_prologue_does_not_grow_with_python_objects_per_environment():
    """256 environments x 8 shards with a scalar control (kappa): ONE equation is built per step and the per-environment
    values travel as an array (the serial prologue was 3.3 ms = 17 % of a 15.8 ms GPU step, i.e. at most
    6.6x of 8 GPUs; target <= 0.8 ms, measured 0.56 ms in an idle container).  Gates that a loaded container cannot
    flip: one equation construction per step, not 256, and the number of Python calls the calling thread makes."""
    import time

    dom = std_domain(P, 16, 16)
    built = []

    class Counted(P.CahnHilliard2DPeriodic):
        def __post_init__(self):
            built.append(1)
            super().__post_init__()

    kw = _kw(dom)
    kw["equation_type"] = Counted
    env = P.VectorPDEEnv(256, **kw, engines=[_NullEngine() for _ in range(8)], device_reward="var", device_observation=(0.0, 1.0))
    env.reset(seed=0)
    acts = [b % 3 for b in range(256)]
    for _ in range(3):
        env.step(acts)
    built.clear()
    # load-independent measure of the serial prologue: Python-level calls made by the CALLING thread per step (the
    # shard threads' work runs beside it).  Round 3: ~10 700 per step at 256 environments (one dataclass construction,
    # two closure traces and three _engine_problem() dicts per environment); now ~1 500, independent of the device count.
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        env.step(acts)
    pr.disable()
    calls_per_step = pstats.Stats(pr).total_calls / 10
    assert len(built) == 10, len(built)  # one equation per step
    assert calls_per_step < 3000, calls_per_step
    # what the environments were handed: per-environment kappas, different across the batch
    k = np.concatenate([sh.engine.kappa_env for sh in env._shards])
    assert len(set(np.round(k, 9))) > 1 and k.shape == (256,)
    # wall time, for the record (printed with -s; 0.56 ms here when the container is otherwise idle, 5.3 ms in round 3)
    t0 = time.perf_counter()
    for _ in range(10):
        env.step(acts)
    print(f"VectorPDEEnv.step, 256 environments x 8 null engines: {1e2 * (time.perf_counter() - t0):.2f} ms per step")
    env.close()


def test_scalar_control_batch_equals_per_environment_equations():
    """the one-equation fast path must hand the engines exactly what 256 separately constructed equations did"""
    from pde_opt_amd.pde_env import _ScalarControlBatch

    dom = std_domain(P, 16, 16)
    eq0 = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    batch = _ScalarControlBatch(eq0, "kappa", [0.002, 0.003, 0.001])
    assert len(batch) == 3 and batch[0] is eq0
    for b, k in enumerate([0.002, 0.003, 0.001]):
        want = P.CahnHilliard2DPeriodic(dom, k, MU["regsol"], MOB["c1mc"])
        got = batch[b]
        assert got.kappa == k and got._engine_problem()["kappa"] == want._engine_problem()["kappa"]
        assert got._engine_problem()["mu"].coef == want._engine_problem()["mu"].coef
        np.testing.assert_array_equal(np.asarray(got.fourier_symbol), np.asarray(want.fourier_symbol))
        assert got.rhs.__self__ is got  # the instance-bound rhs follows the clone, not the template
    sub = batch[1:]
    assert len(sub) == 2 and sub[0].kappa == 0.003 and list(sub.values) == [0.003, 0.001]
    assert [e.kappa for e in batch] == [0.002, 0.003, 0.001]
    with pytest.raises(ValueError, match="plain scalar"):
        eq0._clone_with_scalar("mu", 1.0)


def test_bench_single_process_mode_on_engine_doubles():
    """bench.py --gpus N --single-process: the one-process-many-devices product path has a bench entry;
    here its control flow on two oracle-backed engine doubles -- JSON line, per-device clocks, parity spot of the first
    and last environment against the C oracle"""
    import argparse
    import json
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    args = argparse.Namespace(workload="ch_rk4_64_f32_small", batch_per_gpu=1, gpus=2, steps=1, warmup=0, no_parity_spot=False)
    import contextlib
    import io

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        line = bench.run_single_process(args, P, engines=[OracleEngine(), OracleEngine()])
    printed = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert printed["n_gpus"] == 2 and printed["config"]["envs_total"] == 2 and printed["scaling"] == "weak"
    assert len(printed["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in printed["per_rank_ms_per_step"])
    assert printed["shard_bounds"] == [[0, 1], [1, 2]]
    assert line["parity_spot_ok"] and line["parity_spot_rel_err"] < 1e-4, line
