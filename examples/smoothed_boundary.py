"""notebooks/smooth_boundary.ipynb: a binary disc relaxed into a smooth level set psi by `Shape` (itself a GPU
solve: Allen-Cahn smoothing without curvature flow, Tsit5 + PID), the lowest graph-Laplacian modes of the mask,
then Cahn-Hilliard inside the shape with adaptive Tsit5 + PID and a time-dependent contact angle theta(t)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))  # run from a checkout

import numpy as np

from pde_opt_amd import CahnHilliard2DSmoothedBoundary, Domain, PIDController, SaveAt, Shape, Tsit5, diffeqsolve

quick = "--quick" in sys.argv
Nx = Ny = 100
binary_mask = np.zeros((Nx, Ny))
y, x = np.ogrid[:Nx, :Ny]
binary_mask[np.sqrt((x - 50) ** 2 + (y - 50) ** 2) <= 30] = 1
shape = Shape(binary=binary_mask, dx=(1.0, 1.0), smooth_epsilon=3.0, smooth_curvature=0.008, smooth_dt=0.01,
              smooth_tf=10.0 if quick else 100.0)
psi = shape.smooth
print("psi in [%.3g, %.3g], %d cells of diffuse interface" % (psi.min(), psi.max(), np.sum((psi > 0.01) & (psi < 0.98))))
assert psi.min() == 0.001 and psi.max() == 1.0 and abs(psi.sum() - binary_mask.sum()) < 0.1 * binary_mask.sum()
shape.get_shape_modes(N=6 if quick else 20)
print("lowest mask modes:", np.round(shape.shape_basis_evals[:6], 5))
domain = Domain((Nx, Ny), ((0.0, float(Nx)), (0.0, float(Ny))), "dimensionless", shape)

kappa = 1.0
f = lambda c: c * np.log(c) + (1.0 - c) * np.log(1.0 - c) + 3.0 * c * (1.0 - c) + 0.059  # noqa: E731
mu = lambda c: np.log(c / (1.0 - c)) + 3.0 * (1.0 - 2.0 * c)  # noqa: E731
D = lambda c: (1.0 - c) * c  # noqa: E731

eq = CahnHilliard2DSmoothedBoundary(domain, kappa, f, mu, D, lambda t: np.pi / 2.0, lambda t: 0.0, derivs="fd")
u0 = 0.9 * np.ones((Nx, Ny))
u0[:, :50] = 0.1
t_final = 2.0 if quick else 50.0
solution = diffeqsolve(eq, Tsit5(), t0=0.0, t1=t_final, dt0=1e-3, y0=u0,
                       stepsize_controller=PIDController(rtol=1e-4, atol=1e-6),
                       saveat=SaveAt(ts=np.linspace(0.0, t_final, 20)), max_steps=1000000)
print(solution.stats)  # kernel: tsit5_coop<...>: the whole adaptive solve in one launch, several workgroups per environment
m0, m1 = np.sum(psi * solution.ys[0]), np.sum(psi * solution.ys[-1])
print("psi-weighted mass at t0 / t1:", m0, m1)
assert abs(m1 - m0) < 1e-5 * abs(m0)  # no boundary flux: the mass inside the shape is conserved


def theta(t):  # the notebook's quadratic ramp of the contact angle
    return 34.9065850398866 * (t / t_final * 0.3) ** 2 - 10.4719755119660 * (t / t_final * 0.3) + np.pi / 2


eq2 = CahnHilliard2DSmoothedBoundary(domain, kappa, f, mu, D, theta, lambda t: 0.0, derivs="fd")
solution2 = diffeqsolve(eq2, Tsit5(), t0=0.0, t1=t_final, dt0=1e-3, y0=solution.ys[-1],
                        stepsize_controller=PIDController(rtol=1e-4, atol=1e-6),
                        saveat=SaveAt(ts=np.linspace(0.0, t_final, 20)), max_steps=1000000)
print(solution2.stats)
assert np.isfinite(solution2.ys).all()
print("mean inside the shape:", np.sum(psi * solution2.ys[-1]) / np.sum(psi))
