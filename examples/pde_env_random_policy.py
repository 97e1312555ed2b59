"""notebooks/test_pde_env.ipynb: one episode of a Cahn-Hilliard control environment under a random
policy (kappa is the control), first as a single PDEEnv, then 16 of them as one VectorPDEEnv whose
rewards are reduced on the GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))  # run from a checkout
import time

import numpy as np

from pde_opt_amd import RK4, CahnHilliard2DPeriodic, Domain, PDEEnv, VectorPDEEnv

quick = "--quick" in sys.argv
Nx = Ny = 64 if quick else 128
domain = Domain((Nx, Ny), ((-0.005 * Nx, 0.005 * Nx), (-0.005 * Ny, 0.005 * Ny)), "dimensionless")


def reset_func(domain, seed=0):
    return 0.5 * np.ones(domain.points) + 0.01 * np.random.default_rng(seed).standard_normal(domain.points)


params = dict(
    equation_type=CahnHilliard2DPeriodic,
    domain=domain,
    solver_type=RK4,
    end_time=(5 if quick else 30) * 2e-5,
    step_dt=2e-5,
    numeric_dt=2e-7,
    state_to_observation_func=lambda s: np.clip(s * 255, 0, 255).astype(np.uint8)[None],
    reward_function=lambda x: np.var(x),
    reset_func=reset_func,
    reset_control_value=0.002,
    update_control_value=lambda offset, old: float(np.clip(old + offset, 0.0005, 0.004)),
    update_control_parameter=lambda old, new: new,
    action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -0.0002, 1: 0.0, 2: 0.0002}},
    static_equation_parameters={"mu": lambda c: np.log(c / (1.0 - c)) + 3.0 * (1.0 - 2.0 * c), "D": lambda c: (1.0 - c) * c},
    control_equation_parameter_name="kappa",
    solver_parameters={},
)

env = PDEEnv(**params)
print("Action space:", env.action_space, " Observation space:", env.observation_space)
observation, info = env.reset(seed=0)
episode_over, total_reward, iters = False, 0.0, 0
rng = np.random.default_rng(1)
t0 = time.perf_counter()
while not episode_over:
    action = int(rng.integers(3))
    observation, reward, terminated, truncated, info = env.step(action)
    total_reward += reward
    episode_over = terminated or truncated
    iters += 1
print(f"episode finished after {iters} steps ({time.perf_counter() - t0:.2f} s); total reward {total_reward:.6f}")
assert observation.shape == (1, Nx, Ny) and observation.dtype == np.uint8 and iters >= 5

venv = VectorPDEEnv(16, **params, device_reward="var", fetch_observations=False)
venv.reset(seed=0)
t0 = time.perf_counter()
done, steps = False, 0
while not done:
    _, rewards, terminated, truncated, _ = venv.step(rng.integers(3, size=16))
    done = bool(np.all(terminated | truncated))
    steps += 1
print(f"16 environments x {steps} steps in {time.perf_counter() - t0:.2f} s; last rewards {np.round(rewards[:4], 8)}")
