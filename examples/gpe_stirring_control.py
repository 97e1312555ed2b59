"""Stirring a Bose-Einstein condensate with a moving laser spot -- the RL control the reference's
``GPE2DTSControl.lights(t, x, y)`` + ``PDEEnv`` are built for (pde_opt/numerics/equations/gross_pitaevskii.py:43,61;
pde_opt/pde_env.py:277-303; the reward uses pde_opt/rl_utils.py:19-84 ``detect_vortices``).

The control value is the spot's position on a circle (an angle); an action moves it, and during the environment
step the spot travels from the old to the new position: ``update_control_parameter(old, new)`` returns a callable of
LOCAL time (it restarts at 0 every step, pde_env.py:296-297), which the reference evaluates in every Strang substep
(numerics/solvers.py:109).  ``GaussianSpots`` is such a callable -- it works with the reference unchanged -- and on
the MI355X the split-step kernels evaluate it themselves at every substep's t0, so the time-dependent control
costs no host round trip.  First one ``PDEEnv`` episode, then the same as a ``VectorPDEEnv`` in which every
environment steers its own spot (one batched launch per pass; vortex census on the device).

    python examples/gpe_stirring_control.py [--quick]
"""
import sys
import time

import numpy as np

import pde_opt_amd as P

quick = "--quick" in sys.argv
N = 64 if quick else 256
STEPS = 4 if quick else 40
ENVS = 3 if quick else 32
STEP_DT, NUMERIC_DT, RADIUS = 2e-2, 1e-3, 3.0

dom = P.Domain((N, N), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
X, Y = dom.mesh()


def reset(domain, seed=0):
    rng = np.random.default_rng(seed)
    psi = np.exp(-(X**2 + Y**2) / (2 * 4.0**2)) * np.exp(0.02j * rng.standard_normal(X.shape))
    psi = psi / np.sqrt(np.sum(np.abs(psi) ** 2) * domain.dx[0] ** 2)
    return np.stack([psi.real, psi.imag], axis=-1)


def spot(old_angle, new_angle):
    """the spot moves along the chord from the old to the new position during the step"""
    start = (RADIUS * np.cos(old_angle), RADIUS * np.sin(old_angle))
    end = (RADIUS * np.cos(new_angle), RADIUS * np.sin(new_angle))
    return P.GaussianSpots.moving(40.0, start, end, STEP_DT, 0.8)


def vortex_reward(state):
    from pde_opt_amd.rl_utils import detect_vortices

    return float(detect_vortices(state[..., 0] + 1j * state[..., 1], amp_thresh=1e-4)["num_vortices"])


kw = dict(
    equation_type=P.GPE2DTSControl, domain=dom, solver_type=P.StrangSplitting, end_time=STEPS * STEP_DT, step_dt=STEP_DT,
    numeric_dt=NUMERIC_DT, state_to_observation_func=lambda s: (s[..., 0] ** 2 + s[..., 1] ** 2)[None],
    reward_function=vortex_reward, reset_func=reset, reset_control_value=0.0,
    update_control_value=lambda offset, old: old + offset, update_control_parameter=spot,
    action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -0.4, 1: 0.0, 2: 0.4}},
    static_equation_parameters=dict(k=800.0, e=0.0, trap_factor=1.0, kinetic=True),
    control_equation_parameter_name="lights", solver_parameters={"time_scale": 1.0},
)

env = P.PDEEnv(**kw)
env.reset(seed=0)
t0 = time.perf_counter()
done, total = False, 0.0
while not done:
    obs, reward, done, _, _ = env.step(2)  # keep stirring counter-clockwise
    total += reward
el = time.perf_counter() - t0
norm = float(np.sum(obs) * dom.dx[0] ** 2)
print(f"PDEEnv: {STEPS} steps of {N}^2 ({int(STEP_DT / NUMERIC_DT)} split steps each) in {el:.2f} s; "
      f"vortices at the end {reward:.0f}, norm {norm:.6f}, kernel {env._engine.last_kernel}")
assert abs(norm - 1.0) < 1e-5 and np.isfinite(obs).all()
env.close()

# reward = number of vortices, counted on the device (24 bytes per environment cross PCIe); no field leaves the GPU
venv = P.VectorPDEEnv(ENVS, **kw, fetch_observations=False, device_reward=("vortices", 1e-4, 0.5))
venv.reset(seed=0)
rng = np.random.default_rng(1)
t0 = time.perf_counter()
for _ in range(STEPS):
    actions = rng.integers(0, 3, size=ENVS)
    _, rewards, _, _, _ = venv.step(list(actions))
el = time.perf_counter() - t0
print(f"VectorPDEEnv: {ENVS} environments x {STEPS} steps in {el:.2f} s ({ENVS * STEPS / el:.0f} env-steps/s), each steering "
      f"its own spot; vortices per environment: {rewards.astype(int).tolist()}")
states = venv.states
dens = (states[..., 0].astype(np.float64) ** 2 + states[..., 1].astype(np.float64) ** 2).sum(axis=(1, 2)) * dom.dx[0] ** 2
assert np.allclose(dens, 1.0, atol=1e-5)
venv.close()
print("ok")
