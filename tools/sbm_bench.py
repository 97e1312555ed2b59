"""Smoothed-boundary RHS / RK4 timing on one field (docs notebook scale)."""
import sys, time, types
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pde_opt_amd as P
from pde_opt_amd import _lib as L
from util import MOB, MU, SBM_F, SBM_FLUX, SBM_THETA, sbm_psi

for n in (128, 1024):
    psi = sbm_psi(n, n)
    dom = P.Domain((n, n), ((0.0, float(n)), (0.0, float(n))), "dimensionless", geometry=types.SimpleNamespace(smooth=psi))
    for kind in ("ch", "ac"):
        eq = (P.CahnHilliard2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA, SBM_FLUX) if kind == "ch"
              else P.AllenCahn2DSmoothedBoundary(dom, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA))
        y0 = np.clip(0.5 + 0.1 * np.random.default_rng(0).standard_normal((n, n)), 0.1, 0.9).astype(np.float32)
        eng = P.HipEngine()
        eng.configure(dtype=np.float32, batch=1, **eq._engine_problem()); eq._engine_upload(eng, 0.0); eng.set_state(y0)
        eng.advance(L.INT_RK4, 1e-4, 10, 0.0); eng.sync()
        t0 = time.perf_counter(); eng.advance(L.INT_RK4, 1e-4, 100, 0.0); eng.sync(); el = time.perf_counter() - t0
        print(f"{kind}-sbm {n}^2 fp32: {el / 400 * 1e6:.1f} us per RHS evaluation ({eng.last_kernel})")
        eng.close()
