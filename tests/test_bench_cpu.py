"""bench.py's workload definitions on the CPU: every adaptive workload builds its problem and its parity / CPU-baseline leg
(the same solve driven step by step on the numpy oracle) runs; the 3-D workload's problem has the shape the line reports."""
import numpy as np
import pytest

import bench
import pde_opt_amd as P
from fake_engine import OracleEngine


@pytest.mark.parametrize("name", sorted(bench.ADAPTIVE))
def test_adaptive_workload_problem_and_oracle_leg(name):
    w = bench.ADAPTIVE[name]
    eq, y0 = bench.adaptive_problem(P, name)
    assert y0.shape == (w["n"], w["n"]) and np.all(np.isfinite(y0))
    ctl = P.PIDController(rtol=w["rtol"], atol=w["atol"])
    t1 = w["t1"] / 20  # a prefix: the bench line runs the whole span
    sol = P.diffeqsolve(eq, P.Tsit5(), 0.0, t1, w["dt0"], y0.astype(np.float64), stepsize_controller=ctl, engine=OracleEngine(),
                        saveat=P.SaveAt(ts=np.linspace(0.0, t1, 2)))
    assert sol.stats["num_accepted_steps"] > 5 and np.all(np.isfinite(sol.ys))
    # the notebook's problem: phase boundary at mid-width inside a disc-shaped level set / a steady velocity field
    if name.startswith("ch_sbm"):
        assert eq.psi.shape == y0.shape and 0.0 < eq.psi.min() < 0.01 and eq.psi.max() > 0.99
        assert abs(float(y0[:, : w["n"] // 2].mean()) - 0.1) < 1e-12 and abs(float(y0[:, w["n"] // 2:].mean()) - 0.9) < 1e-12


def test_3d_workload_problem():
    w = dict(bench.WORKLOADS["ch3d_rk4_128_f32"])
    bench.WORKLOADS["_ch3d_small"] = dict(w, n=12)
    try:
        eq, y0, solver = bench.make_problem(P, "_ch3d_small", 2, 0)
    finally:
        del bench.WORKLOADS["_ch3d_small"]
    assert y0.shape == (2, 12, 12, 12) and y0.dtype == np.float32 and len(eq.domain.points) == 3
    assert solver.integrator == P.RK4().integrator
