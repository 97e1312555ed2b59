"""Latency of one env-step (100 RK4 substeps) on the small grids the reference's tests and notebooks use (32^2 ..
256^2): the whole-environment-step kernel (stencil_small.hpp, one launch per pdeopt_advance) against the tiled
stage-pair kernels (two dependent launches per substep) and the multi-workgroup fixed-step kernel (stencil_coop_adaptive.hpp,
MODE 1: several compute units per environment).  usage: python tools/small_grid_bench.py [eq]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P
from pde_opt_amd import _lib as L

substeps = 100
kind = sys.argv[1] if len(sys.argv) > 1 else "ch"
for dtype in (np.float32, np.float64):
    for n in (32, 64, 96, 128, 256):
        for batch in (1, 16, 64, 256, 512):
            if n == 256 and batch > 16:
                continue
            dom = P.Domain((n, n), ((-0.005 * n, 0.005 * n),) * 2, "dimensionless")
            if kind == "ch":
                eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c), lambda c: c * (1 - c))
                dt = 2e-7
            else:
                eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
                dt = 5e-5
            rng = np.random.default_rng(0)
            y0 = np.clip(0.5 + 0.01 * rng.standard_normal((batch, n, n)), 0.05, 0.95).astype(dtype)
            row = []
            for opt in (1, -1, 2, 0):  # one CU per environment; tiled; several CUs per environment; the library's own choice
                eng = P.HipEngine()
                eng.set_small_persist(opt)
                eng.configure(dtype=dtype, batch=batch, **eq._engine_problem())
                eng.set_state(y0)
                for _ in range(3):
                    eng.advance(L.INT_RK4, dt, substeps, 0.0)
                eng.sync()
                reps = 10
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.advance(L.INT_RK4, dt, substeps, 0.0)
                    eng.sync()
                el = (time.perf_counter() - t0) / reps
                row.append((el, eng.last_kernel))
                eng.close()
            (a, ka), (b, kb), (c, kc), (d, kd) = row
            tag = "whole-step" if ka.startswith("small_persist") else "(n/a)"
            multi = f"{c * 1e3:8.3f} ms" if "_coop<" in kc else "   (n/a)   "
            print(f"{kind} {np.dtype(dtype).name} {n:4d}^2 x {batch:3d} envs: {tag} {a * 1e3:8.3f} ms | tiled {b * 1e3:8.3f} ms | several CUs per env {multi} | "
                  f"auto {d * 1e3:8.3f} ms per env-step [{kd}]", flush=True)
