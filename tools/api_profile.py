"""where does VectorPDEEnv.step spend its host time?  (cProfile of 10 steps of the headline workload)"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pde_opt_amd as P
import bench

n, batch = 1024, 32
dom = P.Domain((n, n), ((-5.12, 5.12), (-5.12, 5.12)), "dimensionless")
reset = lambda domain, seed=0: np.clip(0.5 + 0.01 * np.random.default_rng(seed).standard_normal((n, n)), 0.05, 0.95).astype(np.float32)
obs_mode = sys.argv[1] if len(sys.argv) > 1 else "u8"
env = P.VectorPDEEnv(batch, P.CahnHilliard2DPeriodic, dom, P.RK4, end_time=1e9, step_dt=2e-5, numeric_dt=2e-7,
                     state_to_observation_func=lambda s: s, reward_function=lambda s: 0.0, reset_func=reset,
                     reset_control_value=0.002, update_control_value=lambda off, old: old + off,
                     update_control_parameter=lambda old, new: new,
                     action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -1e-5, 1: 0.0, 2: 1e-5}},
                     static_equation_parameters={"mu": bench.REGSOL, "D": bench.C1MC}, control_equation_parameter_name="kappa",
                     solver_parameters={}, device_reward="var",
                     device_observation=(0.0, 1.0) if obs_mode == "u8" else ("probes", [(1, 2), (500, 600)]) if obs_mode == "probes" else None,
                     fetch_observations=False)
env.reset(seed=0)
acts = [b % 3 for b in range(batch)]
for _ in range(2):
    env.step(acts)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    env.step(acts)
pr.disable()
print("obs", obs_mode, "ms per step", (time.perf_counter() - t0) * 100)
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
