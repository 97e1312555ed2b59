"""Where does an adaptive Tsit5 + PID run spend its time? (tests/test_solvers.py:64-104 set-up)"""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
import pde_opt_amd as P

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 1)
dom = P.Domain((nx, ny), ((-0.005 * nx, 0.005 * nx), (-0.005 * ny, 0.005 * ny)), "dimensionless")
eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c), derivs="fd")
u0 = np.ones((nx, ny)); u0[: nx // 2, :] = -1.0
kw = dict(t0=0.0, t1=10.0, dt0=0.00005, y0=u0, saveat=P.SaveAt(ts=np.linspace(0.0, 10.0, 200)),
          stepsize_controller=P.PIDController(rtol=1e-4, atol=1e-6), max_steps=1000000)
P.diffeqsolve(eq, P.Tsit5(), **kw)
t0 = time.perf_counter()
sol = P.diffeqsolve(eq, P.Tsit5(), **kw)
el = time.perf_counter() - t0
n = sol.stats["num_steps"]
print(f"{nx}x{ny}: {n} steps in {el:.3f} s = {el / n * 1e6:.1f} us per step", sol.stats)
pr = cProfile.Profile(); pr.enable(); P.diffeqsolve(eq, P.Tsit5(), **kw); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
