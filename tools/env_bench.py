import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
import pde_opt_amd as P
n = 1024; B = 32
L_ = 0.01 * n
dom = P.Domain((n, n), ((-L_/2, L_/2), (-L_/2, L_/2)), "dimensionless")
def reset(domain, seed=0):
    rng = np.random.default_rng(seed)
    return np.clip(0.5 + 0.01 * rng.standard_normal(domain.points), 0.05, 0.95).astype(np.float32)
kw = dict(equation_type=P.CahnHilliard2DPeriodic, domain=dom, solver_type=P.RK4, end_time=1.0, step_dt=2e-5, numeric_dt=2e-7,
    state_to_observation_func=lambda s: s, reward_function=lambda s: float(np.var(s)), reset_func=reset,
    reset_control_value=0.002, update_control_value=lambda off, old: old + off, update_control_parameter=lambda old, new: new,
    action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -1e-5, 1: 0.0, 2: 1e-5}},
    static_equation_parameters={"mu": lambda c: np.log(c/(1-c)) + 3*(1-2*c), "D": lambda c: c*(1-c)},
    control_equation_parameter_name="kappa", solver_parameters={})
for mode in ("device_reward", "host"):
    venv = P.VectorPDEEnv(B, **kw, device_reward="var" if mode == "device_reward" else None, fetch_observations=(mode == "host"))
    venv.reset(seed=0)
    venv.step([1] * B)
    t0 = time.perf_counter(); K = 5
    for k in range(K):
        obs, rew, term, trunc, info = venv.step([k % 3] * B)
    el = time.perf_counter() - t0
    print(f"VectorPDEEnv[{mode}] {B} envs: {1e3*el/K:.1f} ms/step -> {B*K/el:.0f} env-steps/s")
    venv.close()
env = P.PDEEnv(**kw)
env.reset(seed=0); env.step(1)
t0 = time.perf_counter()
for k in range(5): env.step(k % 3)
el = time.perf_counter() - t0
print(f"PDEEnv single: {1e3*el/5:.1f} ms/step -> {5/el:.0f} env-steps/s")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); env.step(1); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
