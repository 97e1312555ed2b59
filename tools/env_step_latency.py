"""Steady-state latency of PDEEnv.step (pde_env.py:244-317) on a notebook-sized grid: how much of it is the
advance on the GPU and how much host work (control update, D2H of the state, reward / observation callbacks)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pde_opt_amd import RK4, CahnHilliard2DPeriodic, Domain, PDEEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
domain = Domain((N, N), ((-0.005 * N, 0.005 * N),) * 2, "dimensionless")
env = PDEEnv(
    equation_type=CahnHilliard2DPeriodic, domain=domain, solver_type=RK4, end_time=1.0, step_dt=2e-5, numeric_dt=2e-7,
    state_to_observation_func=lambda s: np.clip(s * 255, 0, 255).astype(np.uint8)[None],
    reward_function=lambda x: np.var(x),
    reset_func=lambda d, seed=0: 0.5 + 0.01 * np.random.default_rng(seed).standard_normal(d.points),
    reset_control_value=0.002, update_control_value=lambda off, old: float(np.clip(old + off, 0.0005, 0.004)),
    update_control_parameter=lambda old, new: new,
    action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -0.0002, 1: 0.0, 2: 0.0002}},
    static_equation_parameters={"mu": lambda c: np.log(c / (1.0 - c)) + 3.0 * (1.0 - 2.0 * c), "D": lambda c: (1.0 - c) * c},
    control_equation_parameter_name="kappa", solver_parameters={},
)
env.reset(seed=0)
rng = np.random.default_rng(1)
for _ in range(5):
    env.step(int(rng.integers(3)))
t0 = time.perf_counter()
n = 200
for _ in range(n):
    env.step(int(rng.integers(3)))
el = (time.perf_counter() - t0) / n
print(f"PDEEnv.step {N}^2, 100 RK4 substeps: {el * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    env.step(int(rng.integers(3)))
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(14)
