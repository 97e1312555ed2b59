"""Explicit Euler throughput (the integrator of the reference's PDEEnv.step call site, pde_env.py:293-303)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P
from pde_opt_amd import _lib as L

n, batch, substeps = 1024, 32, 100
dom = P.Domain((n, n), ((-0.005 * n, 0.005 * n),) * 2, "dimensionless")
eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c), lambda c: c * (1 - c))
rng = np.random.default_rng(0)
y0 = np.clip(0.5 + 0.01 * rng.standard_normal((batch, n, n)), 0.05, 0.95).astype(np.float32)
eng = P.HipEngine()
eng.configure(dtype=np.float32, batch=batch, **eq._engine_problem())
eng.set_state(y0)
for integ, name in ((L.INT_EULER, "euler"), (L.INT_RK4, "rk4")):
    eng.advance(integ, 5e-8, substeps, 0.0)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.advance(integ, 5e-8, substeps, 0.0)
    eng.sync()
    el = (time.perf_counter() - t0) / 3
    print(f"{name}: {el * 1e3:.2f} ms per env-step of {substeps} substeps x {batch} envs -> {batch / el:.0f} env-steps/s, "
          f"{el / substeps * 1e6:.1f} us per substep, kernel {eng.last_kernel}")
print("nonfinite", eng.reduce(L.RED_NONFINITE).sum())
