"""Latency of one env-step (100 RK4 substeps) on the small grids the reference's notebooks use
(32^2 .. 256^2, one or a few environments): launch-bound territory.  usage: python tools/small_grid_bench.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P
from pde_opt_amd import _lib as L

substeps = 100
for n in (32, 64, 128, 256):
    for batch in (1, 16):
        dom = P.Domain((n, n), ((-0.005 * n, 0.005 * n),) * 2, "dimensionless")
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c), lambda c: c * (1 - c))
        rng = np.random.default_rng(0)
        y0 = np.clip(0.5 + 0.01 * rng.standard_normal((batch, n, n)), 0.05, 0.95).astype(np.float32)
        eng = P.HipEngine()
        eng.configure(dtype=np.float32, batch=batch, **eq._engine_problem())
        eng.set_state(y0)
        for _ in range(3):
            eng.advance(L.INT_RK4, 2e-7, substeps, 0.0)
        eng.sync()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            eng.advance(L.INT_RK4, 2e-7, substeps, 0.0)
            eng.sync()
        el = (time.perf_counter() - t0) / reps
        print(f"{n:4d}^2 x {batch:2d} envs: {el * 1e3:7.3f} ms per env-step = {el / substeps * 1e6:6.2f} us per substep  ({eng.last_kernel})")
        eng.close()
