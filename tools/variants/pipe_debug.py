import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import pde_opt_amd as P
from pde_opt_amd import _lib as L
from util import MU, MOB, std_domain, white_noise_state
nx, ny, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(17)
dom = std_domain(P, nx, ny)
cubic = len(sys.argv) > 5 and sys.argv[5] == "cubic"
eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["cubic"], MOB["one_plus_sq"]) if cubic else P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
u = white_noise_state(rng, (batch, nx, ny), np.float32, "sym" if cubic else "c")
outs = {}
for mode in (1, 2):
    eng = P.HipEngine(); eng.set_fuse_stages(mode)
    eng.configure(dtype=np.float32, batch=batch, **eq._engine_problem())
    eng.set_env_params(0, kappa=0.002 * (1.0 + 0.1 * np.arange(batch))); eng.set_state(u); eng.advance(L.INT_RK4, 2e-7, steps); outs[mode] = eng.get_state(); print(eng.last_kernel); eng.close()
d = outs[1] != outs[2]
print("increment scale", np.abs(outs[1]-u).max()); print("mismatch cells", d.sum(), "of", d.size, "max abs diff", np.nanmax(np.abs(outs[1] - outs[2])), "nan", np.isnan(outs[2]).sum())
for b in range(batch):
    rows = np.where(d[b].any(axis=1))[0]; cols = np.where(d[b].any(axis=0))[0]
    if len(rows): print("env", b, "rows", rows[:20], "...", rows[-5:], "cols", cols[:20], "...", cols[-5:])
