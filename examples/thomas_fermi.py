"""notebooks/test_thomas_fermi.ipynb: imaginary-time Strang splitting relaxes a Gaussian to the
Thomas-Fermi profile of the trapped condensate."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))  # run from a checkout

import numpy as np

from pde_opt_amd import Domain, GPE2DTSControl, PDEModel, StrangSplitting

quick = "--quick" in sys.argv


def density(psi):
    return np.abs(psi) ** 2


atoms, hbar = 5e5, 1.05e-34
omega = 2 * np.pi * 10
omega_z = np.sqrt(8) * omega
mass, a0 = 3.8175406e-26, 5.29177210903e-11
a_s = 100 * a0
N = 64 if quick else 128
x_s, t_s = np.sqrt(hbar / (mass * omega)), 1 / omega
Lx_ = 150e-6 / x_s
k = 4 * np.pi * a_s * atoms * np.sqrt((mass * omega_z) / (2 * np.pi * hbar))
t_final_, dt_ = (0.03 if quick else 0.1) / t_s, 1e-5 / t_s

domain_ = Domain((N, N), ((-Lx_ / 2, Lx_ / 2), (-Lx_ / 2, Lx_ / 2)), "dimensionless")
X, Y = domain_.mesh()
Psi0_ = np.exp(-(X**2 + Y**2) / (2 * (Lx_ / 6) ** 2)).astype(complex)
Psi0_ /= np.sqrt(np.sum(density(Psi0_)) * domain_.dx[0] ** 2)

pde_model = PDEModel(equation_type=GPE2DTSControl, domain=domain_, solver_type=StrangSplitting)
solution = pde_model.solve(
    parameters={"k": k, "e": 0.0, "lights": lambda t, x, y: 0.0, "trap_factor": 1.0, "kinetic": True},
    y0=np.stack([Psi0_.real, Psi0_.imag], axis=-1),
    ts=np.linspace(0.0, t_final_, 20),
    solver_parameters={"time_scale": -1j},
    dt0=dt_,
)
final = density(solution[-1][..., 0] + 1j * solution[-1][..., 1])

mu = np.sqrt(k / np.pi)  # 2-D Thomas-Fermi chemical potential for unit norm and V = (x^2 + y^2) / 2
tf = np.clip((mu - 0.5 * (X**2 + Y**2)) / k, 0.0, None)
tf *= 1.0 / (tf.sum() * domain_.dx[0] ** 2)
print("norm:", final.sum() * domain_.dx[0] ** 2, " max |n - n_TF|:", np.abs(final - tf).max(), " peak n_TF:", tf.max())
assert abs(final.sum() * domain_.dx[0] ** 2 - 1.0) < 1e-3  # renormalised before the last kinetic half step (solvers.py:111-113)
if not quick:
    np.testing.assert_allclose(final, tf, rtol=1e-3, atol=1e-3)
