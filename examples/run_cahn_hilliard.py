"""notebooks/run_cahn_hilliard.ipynb on the MI355X: spinodal decomposition with the IMEX solver."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))  # run from a checkout
import time

import numpy as np

from pde_opt_amd import CahnHilliard2DPeriodic, Domain, SaveAt, SemiImplicitFourierSpectral, diffeqsolve

quick = "--quick" in sys.argv
Nx, Ny = (64, 64) if quick else (256, 256)
Lx, Ly = 0.01 * Nx, 0.01 * Ny
domain = Domain((Nx, Ny), ((-Lx / 2, Lx / 2), (-Ly / 2, Ly / 2)), "dimensionless")

t_start, t_final, dt = 0.0, (0.002 if quick else 0.2), 0.000001
ts_save = np.linspace(t_start, t_final, 20 if quick else 200)
kappa = 0.002

eq = CahnHilliard2DPeriodic(
    domain,
    kappa,
    lambda c: np.log(c / (1.0 - c)) + 3.0 * (1.0 - 2.0 * c),
    lambda c: (1.0 - c) * c,
    derivs="fd",
)
solver = SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)

u0 = 0.5 * np.ones((Nx, Ny)) + 0.01 * np.random.default_rng(0).standard_normal((Nx, Ny))

t0 = time.perf_counter()
solution = diffeqsolve(eq, solver, t0=t_start, t1=t_final, dt0=dt, y0=u0, saveat=SaveAt(ts=ts_save), max_steps=1000000)
print(solution.stats, f"{time.perf_counter() - t0:.2f} s")
print("mean at t0 / t1:", np.mean(solution.ys[0]), np.mean(solution.ys[-1]))
print("range at t1:", solution.ys[-1].min(), solution.ys[-1].max())
assert abs(np.mean(solution.ys[-1]) - np.mean(solution.ys[0])) < 1e-9  # conservative dynamics
if not quick:
    assert np.std(solution.ys[-1]) > np.std(solution.ys[0])  # spinodal decomposition has set in
