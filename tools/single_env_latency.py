"""Latency of ONE small environment through the Python API -- the regime of the reference's own tests and notebooks
(pde_env.py:244-317 with 32^2 ... 128^2 grids): PDEEnv.step (control update, equation rebuild, one pdeopt_advance of 100
RK4 substeps, host reward + observation of the full field) and PDEModel.solve with Tsit5 + PIDController.
usage: python tools/single_env_latency.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P

REGSOL = lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c)
C1MC = lambda c: c * (1 - c)
for n in (32, 64, 128):
    L_ = 0.01 * n
    dom = P.Domain((n, n), ((-L_ / 2, L_ / 2), (-L_ / 2, L_ / 2)), "dimensionless")

    def reset(domain, seed=0):
        return np.clip(0.5 + 0.01 * np.random.default_rng(seed).standard_normal((n, n)), 0.05, 0.95).astype(np.float32)

    env = P.PDEEnv(P.CahnHilliard2DPeriodic, dom, P.RK4, end_time=1e9, step_dt=2e-7 * 100, numeric_dt=2e-7,
                   state_to_observation_func=lambda s_: s_, reward_function=lambda s_: float(np.var(s_)), reset_func=reset,
                   reset_control_value=0.002, update_control_value=lambda off, old: old + off,
                   update_control_parameter=lambda old, new: new,
                   action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -1e-5, 1: 0.0, 2: 1e-5}},
                   static_equation_parameters={"mu": REGSOL, "D": C1MC}, control_equation_parameter_name="kappa",
                   solver_parameters={})
    env.reset(seed=0)
    for _ in range(5):
        env.step(1)
    t0 = time.perf_counter()
    reps = 50
    for i in range(reps):
        env.step(i % 3)
    el = (time.perf_counter() - t0) / reps
    print(f"PDEEnv.step, one CH {n}^2 fp32 environment, 100 RK4 substeps, full field back to the host: {el * 1e3:.3f} ms per step "
          f"= {1 / el:.0f} env-steps/s", flush=True)
    env.close() if hasattr(env, "close") else None

for n in (256,):
    dom = P.Domain((n, 1), ((-1.28, 1.28), (-0.005, 0.005)), "dimensionless")
    model = P.PDEModel(P.AllenCahn2DPeriodic, dom, P.Tsit5)
    u0 = np.ones((n, 1))
    u0[: n // 2] = -1.0
    ts = np.linspace(0.0, 10.0, 200)
    params = dict(kappa=0.002, mu=lambda c: c**3 - c, R=lambda c: np.ones_like(c), derivs="fd")
    kw = dict(dt0=5e-5, stepsize_controller=P.PIDController(rtol=1e-4, atol=1e-6), max_steps=1000000)
    model.solve(params, u0, ts, **kw)
    t0 = time.perf_counter()
    for _ in range(10):
        ys = model.solve(params, u0, ts, **kw)
    el = (time.perf_counter() - t0) / 10
    print(f"PDEModel.solve, the reference's test_1d_allen_cahn_pde_model (256 x 1, Tsit5 + PID, t = 0..10, 200 save points): "
          f"{el * 1e3:.2f} ms per solve", flush=True)
