"""Where does VectorPDEEnv.step spend its time beside pdeopt_advance?  cProfile of 10 steps of the headline workload
through the API (device reward + frames left on the GPU).  usage: python tools/api_profile.py"""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P

n, batch = 1024, 32
L_ = 0.01 * n
dom = P.Domain((n, n), ((-L_ / 2, L_ / 2), (-L_ / 2, L_ / 2)), "dimensionless")
REGSOL = lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c)
C1MC = lambda c: c * (1 - c)

def reset(domain, seed=0):
    rng = np.random.default_rng(seed)
    return np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95).astype(np.float32)

env = P.VectorPDEEnv(
    batch, P.CahnHilliard2DPeriodic, dom, P.RK4, end_time=1e9, step_dt=2e-7 * 100, numeric_dt=2e-7,
    state_to_observation_func=lambda s_: s_, reward_function=lambda s_: 0.0, reset_func=reset, reset_control_value=0.002,
    update_control_value=lambda off, old: old + off, update_control_parameter=lambda old, new: new,
    action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -1e-5, 1: 0.0, 2: 1e-5}},
    static_equation_parameters={"mu": REGSOL, "D": C1MC}, control_equation_parameter_name="kappa",
    solver_parameters={}, device=0, device_reward="var", device_observation=(0.0, 1.0), observations_on_device=True)
env.reset(seed=0)
actions = [(b % 3) for b in range(batch)]
for _ in range(3):
    env.step(actions)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    env.step(actions)
pr.disable()
el = time.perf_counter() - t0
print(f"{el / 10 * 1e3:.2f} ms per step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
