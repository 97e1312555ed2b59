"""Adaptive Tsit5 + PID solves on LDS-resident grids: the in-kernel controller (pdeopt_tsit5_solve_small, one launch
per solve) against the host-driven loop (pdeopt_tsit5_trial / commit, one device->host read per trial step).
usage: python tools/adaptive_bench.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P

for kind, n, batch, dtype in (("ac", (256, 1), 1, np.float32), ("ac", (64, 64), 1, np.float32), ("ch", (32, 32), 1, np.float32),
                              ("ch", (64, 64), 1, np.float32), ("ch", (64, 64), 1, np.float64), ("ch", (64, 128), 1, np.float32),
                              ("ch", (64, 64), 64, np.float32), ("ch", (64, 64), 256, np.float32)):
    nx, ny = n
    dom = P.Domain((nx, ny), ((-0.005 * nx, 0.005 * nx), (-0.005 * ny, 0.005 * ny)), "dimensionless")
    rng = np.random.default_rng(0)
    if kind == "ch":
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c), lambda c: c * (1 - c))
        y0 = np.clip(0.5 + 0.05 * rng.standard_normal((batch, nx, ny)), 0.05, 0.95).astype(dtype)
        t1, dt0 = 1e-4, 1e-7
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
        y0 = (0.3 * rng.standard_normal((batch, nx, ny))).astype(dtype)
        t1, dt0 = 1.0, 1e-4
    ctl = P.PIDController(rtol=1e-4, atol=1e-6, per_environment=batch > 1)
    row = []
    for opt in (0, -1):
        if opt == -1 and batch > 64:
            row.append((float("nan"), 0, "-"))
            continue
        eng = P.HipEngine()
        eng.set_small_persist(opt)
        arg = y0 if batch > 1 else y0[0]
        P.diffeqsolve(eq, P.Tsit5(), 0.0, t1 / 20, dt0, arg, stepsize_controller=ctl, engine=eng)  # warm
        t = time.perf_counter()
        sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, arg, stepsize_controller=ctl, engine=eng)
        el = time.perf_counter() - t
        row.append((el, sol.stats["num_steps"], sol.stats["kernel"]))
        eng.close()
    (a, na, ka), (b, nb, kb) = row
    print(f"{kind} {np.dtype(dtype).name} {nx}x{ny} x {batch:3d} envs: in-kernel {a * 1e3:8.2f} ms / {na} trial steps = {a / na * 1e6:6.1f} us per step | "
          f"host-driven {b * 1e3:8.2f} ms / {nb} = {b / max(nb, 1) * 1e6:6.1f} us per step -> x{b / a:5.1f}   [{ka} | {kb}]", flush=True)
