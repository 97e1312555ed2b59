"""PDEEnv: the reference's Gymnasium environment (pde_opt/pde_env.py:22-317) on the HIP engine.

Same constructor (16 arguments, same names and order), same ``reset(seed, options)`` /
``step(action)`` contract, same attributes (``_state``, ``_time``, ``_control_value``,
``observation_space``, ``action_space``) and the same quirks that are semantics rather than
implementation: integration time restarts at 0 every environment step, the
observation space is declared ``Box(0, 255, (1, *points), uint8)`` whatever the observation
function returns, parameters may change every step.

What is NOT reproduced is the cost model: upstream rebuilds the equation, the solver and a fresh
``jax.jit`` closure per step (:286-294).  Here the equation object is rebuilt only when its
parameters change (cheap: host dataclass) and ``diffeqsolve`` is one ``pdeopt_advance`` call.

``VectorPDEEnv`` (new) advances B independent episodes in lock step in one batched kernel launch
per Runge-Kutta stage; per-environment control parameters travel with the environment.
"""

from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Sequence, Type

import numpy as np

from . import _lib as L
from .engine import HipEngine
from .integrate import diffeqsolve
from .numerics import domains
from .numerics.equations import BaseEquation
from .numerics.solvers import SaveAt
from .sharding import shard_envs
from .spaces import HAVE_GYMNASIUM, Box, Discrete, EnvBase
from .utils import check_equation_solver_compatibility, prepare_solver_params

if HAVE_GYMNASIUM:  # pragma: no cover
    from gymnasium.envs.registration import register, registry

    if "PDEEnv-v0" not in registry:
        register(id="PDEEnv-v0", entry_point="pde_opt_amd.pde_env:PDEEnv")


class PDEEnv(EnvBase):
    """Reinforcement-learning environment that controls one parameter of a PDE."""

    def __init__(
        self,
        equation_type: Type[BaseEquation],
        domain: domains.Domain,
        solver_type,
        end_time: float,
        step_dt: float,
        numeric_dt: float,
        state_to_observation_func: Callable,
        reward_function: Callable,
        reset_func: Callable,
        reset_control_value,
        update_control_value: Callable,
        update_control_parameter: Callable,
        action_space_config: Dict[str, Any],
        static_equation_parameters: Dict[str, Any],
        control_equation_parameter_name: str,
        solver_parameters: Dict[str, Any],
        device: int = 0,
        engine=None,
    ):
        if HAVE_GYMNASIUM:  # pragma: no cover
            super().__init__()
        self.equation_type = equation_type
        self.domain = domain
        self.solver_type = solver_type
        check_equation_solver_compatibility(self.solver_type, self.equation_type)

        self.end_time = end_time
        self.step_dt = step_dt
        self.numeric_dt = numeric_dt
        self.reward_function = reward_function
        self.reset_func = reset_func
        self.state_to_observation_func = state_to_observation_func

        self.observation_space = Box(low=0.0, high=255.0, shape=(1, *self.domain.points), dtype=np.uint8)
        self._setup_action_space(action_space_config)

        self.reset_control_value = reset_control_value
        self.update_control_value = update_control_value
        self.update_control_parameter = update_control_parameter
        self.static_equation_parameters = static_equation_parameters
        self.control_equation_parameter_name = control_equation_parameter_name
        self.solver_parameters = solver_parameters

        # raises HipUnavailableError without a GPU / library (`engine`: a caller-owned HipEngine, e.g. one
        # bound to a stream; the CPU test-suite injects its oracle-backed double here)
        self._engine = engine if engine is not None else HipEngine(device)
        self._state = None
        self._time = 0.0
        self._control_value = reset_control_value

    # ------------------------------------------------------------------------------------
    def _setup_action_space(self, config: Dict[str, Any]):
        if config.get("type", "continuous") == "discrete":
            self.action_space = Discrete(config.get("num_actions", 5))
            self._action_to_direction = config.get("action_mapping", {})
        else:
            self.action_space = Box(
                low=config.get("low", -1.0), high=config.get("high", 1.0), shape=config.get("shape", (2,))
            )
            self._action_to_direction = None

    def _get_obs(self):
        return self.state_to_observation_func(self._state)

    def _get_info(self):
        return {}

    def _terminate(self):
        return self._time >= self.end_time

    # ------------------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        if seed is not None:
            self._state = np.asarray(self.reset_func(self.domain, seed=seed))
        else:
            self._state = np.asarray(self.reset_func(self.domain))
        self._time = 0.0
        self._control_value = self.reset_control_value
        return self._get_obs(), self._get_info()

    def step(self, action):
        offset = action if not self._action_to_direction else self._action_to_direction[action]
        old_control_value = self._control_value
        self._control_value = self.update_control_value(offset, old_control_value)
        control_parameter = self.update_control_parameter(old_control_value, self._control_value)

        params = {**self.static_equation_parameters, self.control_equation_parameter_name: control_parameter}
        eq = self.equation_type(domain=self.domain, **params)
        solver = self.solver_type(**prepare_solver_params(self.solver_type, self.solver_parameters, eq))

        # local time restarts at 0 every environment step, as upstream (pde_env.py:296-297)
        self._state = self._integrate(eq, solver)
        self._time += self.step_dt

        obs = self._get_obs()
        reward = self.reward_function(self._state)
        return obs, reward, self._terminate(), False, self._get_info()

    def _integrate(self, eq, solver):
        """``diffeqsolve(..., t0=0, t1=step_dt, dt0=numeric_dt, saveat=SaveAt(t1=True)).ys[-1]`` with
        device buffers, rocFFT plans and allocations kept across steps (same-shape ``configure`` is a
        parameter update).  The state is re-uploaded every step (4 MiB at 1024^2 fp32, ~0.2 ms): the
        user may have edited ``self._state`` in place, and checking would cost more than copying."""
        from .integrate import constant_step_plan

        y = np.asarray(self._state)
        if y.dtype not in (np.float32, np.float64):
            y = y.astype(np.float64)
        eng = self._engine
        eng.configure(dtype=y.dtype, batch=1, **eq._engine_problem())
        eq._engine_upload(eng, 0.0, self.step_dt)  # time-dependent controls see the local time of every substep
        solver.configure_engine(eng, eq)
        eng.set_state(y)
        n_full, rem = constant_step_plan(0.0, self.step_dt, self.numeric_dt)
        if n_full + (1 if rem > 0 else 0) > 1_000_000:
            raise RuntimeError("max_steps=1000000 reached")  # diffrax default throw=True
        if n_full:
            eng.advance(solver.integrator, self.numeric_dt, n_full, 0.0)
        if rem > 0:
            eng.advance(solver.integrator, rem, 1, n_full * self.numeric_dt)
        return eng.get_state()[0]

    def close(self):
        self._engine.close()


class _ScalarControlBatch:
    """The equations of a batch that differ in ONE plain-number field (``BaseEquation._scalar_controls``): one real
    equation + the per-environment values.  ``VectorPDEEnv.step`` used to construct (and closure-trace) one dataclass
    per environment before any device was fed -- a serial prologue of 3.3 ms at 256 environments; the
    hot callers read ``values`` / ``eq0``, anything else may still index or iterate (clones are made on demand)."""

    def __init__(self, eq0, name, values):
        self.eq0, self.name, self.values = eq0, name, values

    def __len__(self):
        return len(self.values)

    def __getitem__(self, i):
        if isinstance(i, slice):
            vals = self.values[i]
            return _ScalarControlBatch(self.eq0 if i.indices(len(self.values))[0] == 0 else
                                       self.eq0._clone_with_scalar(self.name, vals[0]), self.name, vals)
        if i < 0:
            i += len(self.values)
        return self.eq0 if i == 0 else self.eq0._clone_with_scalar(self.name, self.values[i])

    def __iter__(self):
        return (self[i] for i in range(len(self.values)))


class _EnvShard:
    """Environments [lo, hi) of a VectorPDEEnv on ONE engine (one GPU, one HIP stream).  ``step`` runs on the
    shard's own host thread: the ~10^2..10^3 asynchronous kernel launches of an environment step are issued per
    device in parallel (ctypes releases the GIL inside the library) and the devices compute concurrently; nothing
    is exchanged between shards -- environments are independent units."""

    def __init__(self, env: "VectorPDEEnv", engine, lo: int, hi: int):
        self.env, self.engine, self.lo, self.hi = env, engine, lo, hi
        self.n = hi - lo
        self._configured_key = None
        self._obs_buffer = None
        self._y0 = None
        self.state_host = None
        self.last_step_s = 0.0

    def reset(self, y0):
        self._y0 = y0
        self.state_host = y0
        self._configured_key = None

    def _configure(self, eqs):
        eng = self.engine
        eq0 = eqs[0]
        prob = eq0._engine_problem()
        y0 = self._y0 if self._y0.dtype in (np.float32, np.float64) else self._y0.astype(np.float64)
        key = (prob["equation"], y0.dtype.str, prob["nx"], prob["ny"],
               getattr(prob.get("mu"), "kind", None), getattr(prob.get("mu"), "flags", None),
               len(getattr(prob.get("mu"), "coef", ())), getattr(prob.get("mob"), "kind", None),
               getattr(prob.get("mob"), "flags", None), len(getattr(prob.get("mob"), "coef", ())))
        if key != self._configured_key:
            eng.configure(dtype=y0.dtype, batch=self.n, **prob)
            eng.set_state(y0)
            self._configured_key = key
        # per-environment parameter values
        if isinstance(eqs, _ScalarControlBatch):
            # one equation, one scalar per environment: no per-environment Python objects
            kappa = (np.asarray(eqs.values, dtype=np.float64) if eqs.name == "kappa"
                     else np.full(self.n, float(prob.get("kappa", 0.0))))
            mu = np.tile(np.asarray(prob["mu"].coef, dtype=np.float64), (self.n, 1)) if prob.get("mu") is not None else None
            mob = np.tile(np.asarray(prob["mob"].coef, dtype=np.float64), (self.n, 1)) if prob.get("mob") is not None else None
        else:
            probs = [e._engine_problem() for e in eqs]
            kappa = [p.get("kappa", 0.0) for p in probs]
            mu = [p["mu"].coef for p in probs] if prob.get("mu") is not None else None
            mob = [p["mob"].coef for p in probs] if prob.get("mob") is not None else None
        eng.set_env_params(0, kappa=kappa, mu_coef=mu, mob_coef=mob)
        return eq0

    def get_states(self):
        """current states of this shard's environments (the reset states until the first step configures the engine)"""
        if self._configured_key is None:
            return np.array(self._y0, copy=True)
        return self.engine.get_state()

    def step(self, eqs):
        """one environment step of this shard's environments: returns (obs, rewards)"""
        from .integrate import constant_step_plan

        import time

        t_begin = time.perf_counter()
        env, eng = self.env, self.engine
        eq0 = self._configure(eqs)
        type(eq0)._engine_upload_batch(eng, eqs, 0.0, env.step_dt)
        solver = env.solver_type(**prepare_solver_params(env.solver_type, env.solver_parameters, eq0))
        solver.configure_engine(eng, eq0)
        if solver.integrator == L.INT_IMEX:
            eng.set_env_imex_scale(0, env._imex_scales(eqs))
        n_full, rem = constant_step_plan(0.0, env.step_dt, env.numeric_dt)
        if n_full:
            eng.advance(solver.integrator, env.numeric_dt, n_full, 0.0)
        if rem > 0:
            eng.advance(solver.integrator, rem, 1, n_full * env.numeric_dt)
        rewards = None
        if isinstance(env.device_reward, tuple):
            # ("vortices", amp_thresh, tol): rl_utils.detect_vortices' num_vortices per environment, counted on the
            # device (pde_opt/rl_utils.py:19-84): 24 bytes per environment cross PCIe instead of the wavefunction
            amp, tol = (tuple(env.device_reward[1:]) + (0.0, 0.5))[:2]
            counts, _ = eng.detect_vortices(amp_thresh=float(amp), tol=float(tol), want_winding=False)
            rewards = counts[:, 0].astype(np.float64)
        elif env.device_reward is not None:
            rewards = eng.reduce(env._RED[env.device_reward])
        fetched = False
        if env.device_observation is not None:
            if isinstance(env.device_observation[0], str):
                obs = eng.probe(env.device_observation[1])  # (n, n_cells): point sensors
            else:
                lo, hi = env.device_observation
                if env.observations_on_device:
                    obs = eng.observe_u8_device(lo, hi).torch()[:, None]  # (n, 1, nx, ny) uint8 tensor on this GPU
                else:
                    if env.reuse_observation_buffer and self._obs_buffer is None and hasattr(eng, "pinned_empty"):
                        self._obs_buffer = eng.pinned_empty((self.n,) + tuple(env.domain.points), np.uint8)
                    obs = eng.observe_u8(lo, hi, out=self._obs_buffer)[:, None]  # (n, 1, nx, ny), the declared space
        elif env.fetch_observations or env.device_reward is None:
            self.state_host, fetched = eng.get_state(), True
            obs = np.stack([env.state_to_observation_func(s_) for s_ in self.state_host])
        else:
            obs = None
        if env.device_reward is None:
            if not fetched:  # a host reward function sees the full field, as upstream (pde_env.py:309)
                self.state_host = eng.get_state()
            rewards = np.asarray([env.reward_function(s_) for s_ in self.state_host])
        # wall time of this device's share (the reward reduction / state fetch above synchronised its stream)
        self.last_step_s = time.perf_counter() - t_begin
        return obs, rewards


class VectorPDEEnv:
    """B independent PDEEnv episodes advanced in lock step on one GPU -- or sharded over several (new capability).

    Semantics per environment are exactly ``PDEEnv``'s.  ``step(actions)`` takes one action per
    environment and every environment integrates with ITS control parameter: numbers and closure
    *coefficients* (same closure structure across the batch) travel in per-environment tables, fields
    that depend on the control (the GPE potential for ``e`` / ``lights`` / ``trap_factor``, face velocities
    of advection-diffusion) are uploaded per environment, so the whole batch still runs in one launch per
    stage.  Under the IMEX solver a per-environment ``kappa`` also means a per-environment implicit operator
    (``fourier_symbol = kappa (2 pi i k)^4``): the transforms then carry one environment per complex field
    instead of two (``pdeopt_set_env_imex_scale``).  A control the kernels cannot vary inside one batch (see
    ``_per_env_controls`` of the equation class) raises ``ValueError`` as soon as two environments disagree on
    it -- never a silently shared value.

    ``devices=[0, 1, ...]`` (BASELINE.json north_star: "batched episodes shard naturally across the 8 GPUs of one
    node"): the environments are split into contiguous, balanced blocks (``sharding.shard_envs``), one
    ``HipEngine`` and one host thread per device; ``step`` launches every device's share, the devices compute
    concurrently, and the per-environment results are concatenated in environment order.  There is no collective
    and no data exchanged between devices; results per environment do not depend on the number of
    devices (bitwise for the explicit integrators).  ``engines=[...]`` injects caller-owned engines instead (several
    engines may share one GPU; the CPU test-suite passes its oracle-backed doubles).  One process per GPU with
    ``torch.distributed`` is the other way to scale (``bench.py --gpus N``, ``sharding.gather_per_env``).

    ``reward`` / observations: ``reward_function`` and ``state_to_observation_func`` are applied
    per environment on host copies unless ``device_reward`` names an on-device reduction
    (``"var"``, ``"mean"``, ``"min"``, ``"max"``, or ``("vortices", amp_thresh, tol)`` = the number of quantised
    vortices of a GPE state, ``rl_utils.detect_vortices`` on the device), which avoids the D2H of full fields;
    ``device_observation=(lo, hi)`` forms the uint8 image observations of the declared
    observation space on the GPU (1 byte per cell crosses PCIe instead of 4 or 8);
    ``device_observation=("probes", cells)`` returns the state at the listed grid cells instead (sensor-style
    observations: ``(B, n_cells)`` float64, a few numbers per environment).  ``observations_on_device=True``
    leaves the uint8 frames on the GPU and returns them as a zero-copy ``torch.uint8`` CUDA tensor
    ``(B, 1, nx, ny)`` for a policy on the same device (no PCIe at all; the tensor is overwritten by the next
    step) -- with several devices a list of such tensors, one per device, in environment order.
    """

    def __init__(
        self,
        num_envs: int,
        equation_type,
        domain,
        solver_type,
        end_time,
        step_dt,
        numeric_dt,
        state_to_observation_func,
        reward_function,
        reset_func,
        reset_control_value,
        update_control_value,
        update_control_parameter,
        action_space_config,
        static_equation_parameters,
        control_equation_parameter_name,
        solver_parameters,
        device: int = 0,
        device_reward: Optional[str] = None,
        fetch_observations: bool = True,
        device_observation: Optional[tuple] = None,
        engine=None,
        reuse_observation_buffer: bool = False,
        observations_on_device: bool = False,
        devices: Optional[Sequence[int]] = None,
        engines: Optional[Sequence] = None,
    ):
        self.num_envs = int(num_envs)
        self.observations_on_device = bool(observations_on_device)
        if self.observations_on_device and (device_observation is None or isinstance(device_observation[0], str)):
            raise ValueError("observations_on_device needs device_observation=(lo, hi)")
        if device_observation is not None and isinstance(device_observation[0], str) and device_observation[0] != "probes":
            raise ValueError(f"unknown device observation {device_observation[0]!r}")
        if isinstance(device_reward, tuple) and device_reward[0] != "vortices":
            raise ValueError(f"unknown device reward {device_reward[0]!r}")
        if isinstance(device_reward, str) and device_reward not in self._RED:
            raise ValueError(f"unknown device reward {device_reward!r} (one of {sorted(self._RED)} or (\"vortices\", amp, tol))")
        # True: device-formed uint8 frames land in ONE page-locked host array per device that every step overwrites
        # and returns (copy what you keep) -- no 32 MiB allocation + page faults + pageable D2H per step
        self.reuse_observation_buffer = bool(reuse_observation_buffer)
        self.equation_type, self.domain, self.solver_type = equation_type, domain, solver_type
        check_equation_solver_compatibility(solver_type, equation_type)
        self.end_time, self.step_dt, self.numeric_dt = end_time, step_dt, numeric_dt
        self.state_to_observation_func = state_to_observation_func
        self.reward_function = reward_function
        self.reset_func = reset_func
        self.reset_control_value = reset_control_value
        self.update_control_value = update_control_value
        self.update_control_parameter = update_control_parameter
        self.static_equation_parameters = static_equation_parameters
        self.control_equation_parameter_name = control_equation_parameter_name
        self.solver_parameters = solver_parameters
        self.device_reward = device_reward
        self.fetch_observations = fetch_observations
        self.device_observation = device_observation  # (lo, hi): uint8 frames quantised on the GPU
        self.single_observation_space = Box(low=0.0, high=255.0, shape=(1, *domain.points), dtype=np.uint8)
        cfg = action_space_config
        if cfg.get("type", "continuous") == "discrete":
            self.single_action_space = Discrete(cfg.get("num_actions", 5))
            self._action_to_direction = cfg.get("action_mapping", {})
        else:
            self.single_action_space = Box(low=cfg.get("low", -1.0), high=cfg.get("high", 1.0), shape=cfg.get("shape", (2,)))
            self._action_to_direction = None
        # ---- engines: one per device (raises HipUnavailableError without a GPU / the library)
        if engines is not None and (engine is not None or devices is not None):
            raise ValueError("pass engines=[...] or engine= / devices=, not both")
        # engines the caller passed in stay the caller's: close() shuts down only what this constructor created
        self._owns_engines = engines is None and engine is None
        self._closed = False
        if engines is None:
            if engine is not None:
                if devices is not None:
                    raise ValueError("pass engine= (one caller-owned engine) or devices=[...], not both")
                engines = [engine]
            else:
                devs = [int(device)] if devices is None else [int(d) for d in devices]
                if not devs:
                    raise ValueError("devices is empty")
                if len(set(devs)) != len(devs):
                    raise ValueError(f"devices lists a GPU twice: {devs} (several engines on one GPU: engines=[...])")
                engines = [HipEngine(d) for d in devs]
        engines = list(engines)
        if not 1 <= len(engines) <= self.num_envs:
            raise ValueError(f"{len(engines)} engines for {self.num_envs} environments")
        self._shards = []
        for r, eng in enumerate(engines):
            lo, hi = shard_envs(self.num_envs, len(engines), r)
            self._shards.append(_EnvShard(self, eng, lo, hi))
        self._engine = engines[0]  # the single-device handle existing callers know
        # one DEDICATED host thread per device (a single-worker executor each): a device's launches always come
        # from the same thread, whose HIP "current device" therefore never changes
        self._pool = None
        if len(self._shards) > 1:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"pdeopt-dev{r}") for r in range(len(self._shards))]
        self._time = np.zeros(self.num_envs)
        self._control_value = [reset_control_value] * self.num_envs

    _RED = {"mean": L.RED_MEAN, "var": L.RED_VAR, "min": L.RED_MIN, "max": L.RED_MAX}

    @property
    def num_devices(self) -> int:
        return len(self._shards)

    @property
    def last_step_seconds(self):
        """wall time each device spent on its share of the last ``step`` (one entry per device)"""
        return [s.last_step_s for s in self._shards]

    @property
    def shard_bounds(self):
        """[(lo, hi)] environment ranges, one per device"""
        return [(s.lo, s.hi) for s in self._shards]

    def _equation_for(self, control_parameter):
        params = {**self.static_equation_parameters, self.control_equation_parameter_name: control_parameter}
        return self.equation_type(domain=self.domain, **params)

    def _map_shards(self, fn):
        """fn(shard) on every shard, each on its own host thread; results in shard order (exceptions re-raised)"""
        if self._closed:
            raise RuntimeError("this VectorPDEEnv has been closed")
        if self._pool is None:
            return [fn(s_) for s_ in self._shards]  # one shard (several only if no pool could be made): in order, here
        futs = [ex.submit(fn, s_) for ex, s_ in zip(self._pool, self._shards)]
        results, first_error = [], None
        for f in futs:  # wait for EVERY device before raising: no shard is left running behind an exception
            try:
                results.append(f.result())
            except BaseException as e:  # noqa: BLE001
                first_error = first_error or e
        if first_error is not None:
            raise first_error
        return results

    def reset(self, seed: Optional[int] = None, options=None):
        states = []
        for b in range(self.num_envs):
            s = self.reset_func(self.domain, seed=seed + b) if seed is not None else self.reset_func(self.domain)
            states.append(np.asarray(s))
        self._y0 = np.stack(states)
        self._time[:] = 0.0
        self._control_value = [self.reset_control_value] * self.num_envs
        for sh in self._shards:
            sh.reset(self._y0[sh.lo:sh.hi])
        obs = [self.state_to_observation_func(s) for s in self._y0]
        return np.stack(obs), {}

    @staticmethod
    def _same(a, b) -> bool:
        if a is b:
            return True
        if callable(a) or callable(b):
            return False
        try:
            return bool(np.all(np.asarray(a) == np.asarray(b)))
        except Exception:
            return False

    def _check_controls(self, eqs, controls):
        """environments may only disagree on parameters the batched kernels carry per environment -- checked over
        the WHOLE job, so what is accepted does not depend on how many devices the environments are spread over"""
        eq0 = eqs[0]
        if isinstance(eqs, _ScalarControlBatch):
            # one equation with a per-environment number: the closure structure is shared by construction
            if eqs.name not in type(eq0)._per_env_controls and any(v != eqs.values[0] for v in eqs.values):
                raise ValueError(
                    f"{type(eq0).__name__}: the control parameter {eqs.name!r} cannot differ between the environments "
                    f"of one VectorPDEEnv (per-environment controls: {sorted(type(eq0)._per_env_controls)})")
            return
        prob = eq0._engine_problem()
        for e in eqs[1:]:
            p = e._engine_problem()
            for name in ("mu", "mob"):
                a, b = prob.get(name), p.get(name)
                if (a is None) != (b is None) or (a is not None and (a.kind, a.flags, len(a.coef), getattr(a, "source", "")) !=
                                                  (b.kind, b.flags, len(b.coef), getattr(b, "source", ""))):
                    raise ValueError("all environments of a VectorPDEEnv must share the closure structure")
        if all(self._same(c, controls[0]) for c in controls[1:]):
            return
        name = self.control_equation_parameter_name
        if name not in type(eq0)._per_env_controls:
            raise ValueError(
                f"{type(eq0).__name__}: the control parameter {name!r} cannot differ between the environments "
                f"of one VectorPDEEnv (per-environment controls: {sorted(type(eq0)._per_env_controls)})")

    def _imex_scales(self, eqs):
        """sigma_b of ``pdeopt_set_env_imex_scale`` for the environments ``eqs`` of one engine: fourier_symbol =
        kappa (2 pi i k)^4 (cahn_hilliard.py:74), so with ``kappa`` as the per-environment control every environment
        has its own implicit operator = kappa_b / kappa_0 x the engine's first environment's.  Only then: another
        control leaves the operator shared, and a ``fourier_symbol`` passed in ``solver_parameters`` is the caller's,
        not kappa's."""
        ones = [1.0] * len(eqs)
        if self.control_equation_parameter_name != "kappa" or "fourier_symbol" in (self.solver_parameters or {}):
            return ones
        kappas = [float(v) for v in eqs.values] if isinstance(eqs, _ScalarControlBatch) else [float(e.kappa) for e in eqs]
        if all(k == kappas[0] for k in kappas):
            return ones
        if any(not k > 0.0 for k in kappas):
            raise ValueError(f"per-environment kappa under the IMEX solver must be positive, got {kappas}")
        if len(self.domain.points) != 2:
            raise ValueError("per-environment kappa under the IMEX solver needs the fused 2-D FFT passes "
                             f"({type(eqs[0]).__name__} shares one implicit operator across the batch)")
        return [k / kappas[0] for k in kappas]

    def step(self, actions: Sequence):
        if len(actions) != self.num_envs:
            raise ValueError(f"{len(actions)} actions for {self.num_envs} environments")
        controls = []
        for b, action in enumerate(actions):
            offset = action if not self._action_to_direction else self._action_to_direction[action]
            old = self._control_value[b]
            self._control_value[b] = self.update_control_value(offset, old)
            controls.append(self.update_control_parameter(old, self._control_value[b]))
        name = self.control_equation_parameter_name
        if name in getattr(self.equation_type, "_scalar_controls", ()) and all(
                isinstance(c, (int, float, np.integer, np.floating)) and not isinstance(c, bool) for c in controls):
            # the control is a plain number the equation stores as given: ONE equation (its parameters validated, its
            # closures traced once) + the per-environment values as an array
            eqs = _ScalarControlBatch(self._equation_for(controls[0]), name, controls)
        else:
            eqs = [self._equation_for(c) for c in controls]
        self._check_controls(eqs, controls)
        # every device's share on its own host thread; no collective, nothing exchanged
        results = self._map_shards(lambda sh: sh.step(eqs[sh.lo:sh.hi]))
        self._time += self.step_dt
        rewards = np.concatenate([np.asarray(r[1], dtype=np.float64).reshape(-1) for r in results])
        obs_parts = [r[0] for r in results]
        if obs_parts[0] is None:
            obs = None
        elif self.observations_on_device and self.device_observation is not None and not isinstance(self.device_observation[0], str):
            obs = obs_parts[0] if len(obs_parts) == 1 else obs_parts  # device tensors stay on their GPUs
        else:
            obs = obs_parts[0] if len(obs_parts) == 1 else np.concatenate(obs_parts)
        terminated = self._time >= self.end_time
        return obs, rewards, terminated, np.zeros(self.num_envs, dtype=bool), {}

    @property
    def _state_host(self):
        parts = [sh.state_host for sh in self._shards]
        return parts[0] if len(parts) == 1 else np.concatenate(parts)

    @property
    def states(self):
        parts = self._map_shards(lambda sh: sh.get_states())
        return parts[0] if len(parts) == 1 else np.concatenate(parts)

    def close(self):
        """stop the per-device threads and close the engines this environment created (``device=`` / ``devices=``);
        engines passed in through ``engine=`` / ``engines=`` belong to the caller and stay open.  A closed environment
        raises on ``step`` / ``states`` instead of silently serving the first shard."""
        if self._pool is not None:
            for ex in self._pool:
                ex.shutdown(wait=True)
            self._pool = None
        if self._owns_engines and not self._closed:
            for sh in self._shards:
                sh.engine.close()
        self._closed = True
