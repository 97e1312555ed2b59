"""launch-bound cases: eager vs hipGraph replay"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P
from pde_opt_amd import _lib as L

def run(name, eq, y0, integ, dt, n, reps=5):
    for mode in (-1, 1):
        eng = P.HipEngine(); eng.set_graph(mode)
        eng.configure(dtype=y0.dtype, batch=y0.shape[0], **eq._engine_problem()); eq._engine_upload(eng, 0.0)
        eng.set_state(y0)
        eng.advance(integ, dt, n); eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps): eng.advance(integ, dt, n)
        eng.sync(); el = (time.perf_counter() - t0) / reps
        print(f"{name:28s} graph={mode:2d}  {1e3*el:8.3f} ms per {n} substeps  ({1e6*el/n:6.2f} us/substep)  {eng.last_kernel}")
        eng.close()

rng = np.random.default_rng(0)
dom = P.Domain((128, 128), ((-1.28, 1.28), (-1.28, 1.28)), "d")
eq = P.AdvectionDiffusion2D(dom, lambda t, x, y: (0.1 * np.sin(x), 0.05 * np.cos(y)), 0.1)
run("AD 128^2 f64 Euler x1 (cfg 1)", eq, (0.5 + 0.01 * rng.standard_normal((1, 128, 128))), L.INT_EULER, 1e-4, 512)
mu = lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c); D = lambda c: c * (1 - c)
for n_, b in ((128, 1), (256, 1), (512, 1), (1024, 1), (256, 16)):
    dom = P.Domain((n_, n_), ((-0.005 * n_, 0.005 * n_),) * 2, "d")
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, mu, D)
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((b, n_, n_)), 0.05, 0.95).astype(np.float32)
    run(f"CH {n_}^2 f32 RK4 x{b}", eq, y0, L.INT_RK4, 2e-7, 512)
