"""Fixed-step multi-workgroup kernel (stencil_coop_adaptive.hpp, MODE 1): time per 100 RK4 substeps of ONE Cahn-Hilliard
environment for forced tile edges (PDEOPT_COOP_TILE) against the planner's own choice and the tiled kernels.
usage: python tools/coop_fixed_tile_sweep.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P
from pde_opt_amd import _lib as L


def run(n, dtype, opt, tile):
    if tile:
        os.environ["PDEOPT_COOP_TILE"] = str(tile)
    else:
        os.environ.pop("PDEOPT_COOP_TILE", None)
    dom = P.Domain((n, n), ((-0.005 * n, 0.005 * n),) * 2, "dimensionless")
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c), lambda c: c * (1 - c))
    y0 = np.clip(0.5 + 0.01 * np.random.default_rng(0).standard_normal((1, n, n)), 0.05, 0.95).astype(dtype)
    eng = P.HipEngine()
    eng.set_small_persist(opt)
    eng.configure(dtype=dtype, batch=1, **eq._engine_problem())
    eng.set_state(y0)
    try:
        for _ in range(3):
            eng.advance(L.INT_RK4, 2e-7, 100, 0.0)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.advance(L.INT_RK4, 2e-7, 100, 0.0)
            eng.sync()
        el = (time.perf_counter() - t0) / 10
        k = eng.last_kernel
    except Exception as e:  # noqa: BLE001
        el, k = float("nan"), repr(e)[:80]
    eng.close()
    return el, k


for dtype in (np.float32,):
    for n in (128, 192, 256, 320, 384, 512):
        el, k = run(n, dtype, -1, 0)
        print(f"{np.dtype(dtype).name} {n}^2: tiled {el * 1e3:.3f} ms [{k}]", flush=True)
        for tile in (0, 40, 32, 26, 22, 18, 16):
            el, k = run(n, dtype, 2, tile)
            print(f"    tile {tile:2d}: {el * 1e3:.3f} ms [{k}]", flush=True)
