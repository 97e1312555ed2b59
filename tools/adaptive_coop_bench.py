"""The reference's notebook-sized adaptive solves (Tsit5 + PIDController(rtol 1e-4, atol 1e-6)) on the multi-workgroup
in-kernel path (csrc/stencil_coop_adaptive.hpp) against the host-driven trial / commit loop:
  * notebooks/smooth_boundary.ipynb:228: CahnHilliard2DSmoothedBoundary 100^2, kappa 0.002, dx 0.01, theta = pi / 2
    (117 890 steps over t = 0 .. 0.1 upstream; a prefix of that solve here), and its second solve with theta(t)
  * notebooks/run_advection_diffusion.ipynb:84: advection-diffusion 64^2 (8 950 steps upstream)
  * periodic Cahn-Hilliard 100^2 / 128^2 (beyond one compute unit's registers)
usage: python tools/adaptive_coop_bench.py [t1_scale]"""
import sys, time, types
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
only = sys.argv[2] if len(sys.argv) > 2 else ""  # substring of the case name
dtypes = (np.float32,) if len(sys.argv) > 3 and sys.argv[3] == "f32" else (np.float32, np.float64)
F = lambda c: c * np.log(c) + (1.0 - c) * np.log(1.0 - c) + 3.0 * c * (1.0 - c) + 0.059  # noqa: E731
MU = lambda c: np.log(c / (1.0 - c)) + 3.0 * (1.0 - 2.0 * c)  # noqa: E731
D = lambda c: (1.0 - c) * c  # noqa: E731
THETA = lambda t: 34.9065850398866 * t**2 - 10.4719755119660 * t + np.pi / 2  # noqa: E731


def disc_psi(n, radius=20.0, eps=3.0, floor=1e-3):
    """a disc of the notebook's size as a smooth level set (the notebook relaxes a binary disc with Shape: examples/smoothed_boundary.py)"""
    y, x = np.ogrid[:n, :n]
    r = np.sqrt((x - n / 2) ** 2 + (y - n / 2) ** 2)
    return np.maximum(floor, 0.5 * (1.0 + np.tanh((radius - r) / eps)))


def cases():
    n = 100
    dom = P.Domain((n, n), ((-0.5, 0.5), (-0.5, 0.5)), "dimensionless", geometry=types.SimpleNamespace(smooth=disc_psi(n)))
    u0 = 0.9 * np.ones((n, n))
    u0[:, :50] = 0.1
    yield "CH-SBM 100^2 theta=pi/2 (smooth_boundary.ipynb:228)", P.CahnHilliard2DSmoothedBoundary(dom, 0.002, F, MU, D, lambda t: np.pi / 2.0, lambda t: 0.0), u0, 2e-3, 1e-6
    yield "CH-SBM 100^2 theta(t) (smooth_boundary.ipynb:397)", P.CahnHilliard2DSmoothedBoundary(dom, 0.002, F, MU, D, THETA, lambda t: 0.0), u0, 2e-3, 1e-6
    yield "AC-SBM 100^2 theta=pi/3", P.AllenCahn2DSmoothedBoundary(dom, 0.002, F, MU, D, lambda t: np.pi / 3.0), u0, 2.0, 1e-6
    n = 64
    dom = P.Domain((n, n), ((0.0, 0.02 * n), (0.0, 0.02 * n)), "dimensionless")

    def vel(t, x, y):
        g = np.exp(-((x - 0.4) ** 2 + (y - 0.4) ** 2) / (2 * 0.01))
        return -0.1 * (x - 0.4) / 0.01 * g, -0.1 * (y - 0.4) / 0.01 * g

    rng = np.random.default_rng(0)
    yield "advection-diffusion 64^2 (run_advection_diffusion.ipynb:84)", P.AdvectionDiffusion2D(dom, vel, 0.1, time_dependent=False), 0.5 + 0.01 * rng.standard_normal((n, n)), 2.0, 1e-5
    for n in (64, 100, 128, 192, 256, 384, 512):  # up to 200^2 one XCD's L2; beyond: several XCDs, fenced barrier
        dom = P.Domain((n, n), ((-0.005 * n, 0.005 * n),) * 2, "dimensionless")
        yield f"CH periodic {n}^2", P.CahnHilliard2DPeriodic(dom, 0.002, MU, D), np.clip(0.5 + 0.05 * rng.standard_normal((n, n)), 0.05, 0.95), 1e-4, 1e-7


for name, eq, y0, t1, dt0 in cases():
    if only not in name:
        continue
    t1 *= scale
    for dtype in dtypes:
        ctl = P.PIDController(rtol=1e-4, atol=1e-6)
        row = []
        for opt in (2, 0, -1):  # multi-workgroup kernel; the library's own choice; host-driven
            eng = P.HipEngine()
            eng.set_small_persist(opt)
            y = y0.astype(dtype)
            P.diffeqsolve(eq, P.Tsit5(), 0.0, t1 / 50, dt0, y, stepsize_controller=ctl, engine=eng)  # warm
            t = time.perf_counter()
            sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, dt0, y, stepsize_controller=ctl, engine=eng, saveat=P.SaveAt(ts=np.linspace(0.0, t1, 20)))
            el = time.perf_counter() - t
            row.append((el, sol.stats["num_steps"], sol.stats["kernel"], sol.ys[-1].astype(np.float64)))
            eng.close()
        (a, na, ka, ya), (c, nc, kc, _), (b, nb, kb, yb) = row
        dev = float(np.max(np.abs(ya - yb)))
        print(f"{name} {np.dtype(dtype).name}: multi-workgroup {a * 1e3:8.2f} ms / {na} trial steps = {a / na * 1e6:6.2f} us per step [{ka}] | "
              f"auto {c / nc * 1e6:6.2f} us per step [{kc}] | host-driven {b / max(nb, 1) * 1e6:6.1f} us per step ({nb} steps) -> x{(b / nb) / (a / na):5.1f}; "
              f"max |multi - host| at t1 = {dev:.2e}", flush=True)
