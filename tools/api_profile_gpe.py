"""host-time profile of VectorPDEEnv.step for the GPE stirring example (32 envs x 256^2)"""
import cProfile, os, pstats, sys, time, runpy
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["x", "--quick"]
import pde_opt_amd as P
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples", "gpe_stirring_control.py"))
kw = dict(ns["kw"])
N = 256
dom = P.Domain((N, N), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
X, Y = dom.mesh()
def reset(domain, seed=0):
    psi = np.exp(-(X**2 + Y**2) / 32.0).astype(complex)
    psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * domain.dx[0] ** 2)
    return np.stack([psi.real, psi.imag], axis=-1)
kw.update(domain=dom, reset_func=reset)
venv = P.VectorPDEEnv(32, **kw, fetch_observations=False, device_reward=None)
venv.fetch_observations = False
venv.device_reward = "mean"  # GPE: reduce works on the (re, im) pairs; only to keep fields on the device
venv.reset(seed=0)
acts = [b % 3 for b in range(32)]
for _ in range(2):
    venv.step(acts)
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for _ in range(10):
    venv.step(acts)
pr.disable()
print("ms per step", (time.perf_counter() - t0) * 100)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
