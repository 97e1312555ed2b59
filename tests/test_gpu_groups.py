"""The environment-group path (the one bench.py times) against the oracle, at the sizes it is timed at.

Explicit, IMEX and Strang pipelines advance a large batch group by group of environments sized for the
256 MiB Infinity Cache (csrc/stencil.hip advance_explicit, csrc/strang_fused.hip): every pointer of a
launch is offset by the group's first environment.  Two kinds of checks:

  * grouped == ungrouped BITWISE on small grids with odd batches and a short last group, for every
    pipeline that has a group loop (environments are independent, so grouping may not change a bit);
  * the BASELINE configurations at their full per-GPU batch (configs 2, 3-RK4, 3-IMEX, 4), first / last
    environment of every group against the CPU oracle (oracle/c_oracle.c for RK4, oracle/np_oracle.py for
    the spectral integrators) on the same seeded inputs.
"""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import c_oracle as CO
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import MOB, MU, rel_l2, std_domain, white_noise_state

pytestmark = pytest.mark.gpu


def _advance(eq, solver, y0, integ, dt, n, group, **opts):
    eng = P.HipEngine()
    eng.set_group_envs(group)
    for k, v in opts.items():
        getattr(eng, "set_" + k)(v)
    eng.configure(dtype=y0.dtype, batch=y0.shape[0], **eq._engine_problem())
    eq._engine_upload(eng, 0.0)
    if solver is not None:
        solver.configure_engine(eng, eq)
    eng.set_state(y0)
    eng.advance(integ, dt, n)
    out, groups, kernel = eng.get_state(), eng.last_groups(), eng.last_kernel
    eng.close()
    return out, groups, kernel


# ----------------------------------------------------------------------------- grouped == ungrouped
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", ["ch_rk4_quad", "ch_rk4_pair", "ch_rk4_stage", "ch_rk4_generic", "ch_euler_pair", "ch_euler_odd",
                                  "ac_rk4_quad", "ac_rk4_pair"])
@pytest.mark.parametrize("shape,batch,group", [((64, 128), 5, 2), ((48, 40), 7, 3), ((128, 128), 4, 3)])
def test_explicit_grouped_equals_ungrouped_bitwise(dtype, case, shape, batch, group):
    rng = np.random.default_rng(11)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    if case.startswith("ch"):
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        u = white_noise_state(rng, (batch, nx, ny), dtype, "c")
        dt = 2e-7
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one_plus_sq"])
        u = white_noise_state(rng, (batch, nx, ny), dtype, "sym")
        dt = 5e-5
    opts, integ, n = {}, L.INT_RK4, 6
    if case == "ch_rk4_stage":
        opts["fuse_stages"] = -1
    elif case == "ch_rk4_generic":
        opts["kernel_path"] = L.PATH_GENERIC
    elif case == "ch_euler_pair":
        integ = L.INT_EULER
    elif case == "ch_euler_odd":
        integ, n = L.INT_EULER, 7  # pairs + one single trailing substep
    elif case in ("ac_rk4_pair", "ch_rk4_pair"):
        opts["fuse_stages"] = 1
    whole, g1, k1 = _advance(eq, None, u, integ, dt, n, -1, graph=-1, **opts)
    parts, g2, k2 = _advance(eq, None, u, integ, dt, n, group, graph=-1, **opts)
    assert g1 == 1 and g2 == -(-batch // group), (g1, g2)
    assert k1 == k2
    if case == "ch_rk4_pair":
        assert "stage_pair" in k1, k1
    if case == "ch_rk4_quad" and dtype is np.float32 and nx % 32 == 0 and ny % 128 == 0:
        assert "rk4_quad" in k1, k1  # the whole-substep kernel where its 32 x 128 tiles divide the grid
    if case == "ac_rk4_quad" and dtype is np.float32:
        assert "rk4_quad" in k1, k1
    assert np.isfinite(whole).all() and np.any(whole != u)
    np.testing.assert_array_equal(parts, whole)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", ["ch_rk4_pair", "ch_rk4_stage", "ch_euler_pair", "ch_euler_odd", "ac_rk4_quad"])
@pytest.mark.parametrize("shape,batch,group", [((64, 128), 5, 2), ((48, 40), 7, 3), ((128, 128), 4, 1)])
def test_two_groups_side_by_side_equal_one_at_a_time_bitwise(dtype, case, shape, batch, group):
    """PDEOPT_OPT_GROUP_STREAMS: two groups in flight on two HIP streams (the schedule bench.py's headline runs) --
    the same kernels on the same windows in another order; an odd group count leaves the last one alone"""
    rng = np.random.default_rng(12)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    if case.startswith("ch"):
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        u = white_noise_state(rng, (batch, nx, ny), dtype, "c")
        dt = 2e-7
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
        u = white_noise_state(rng, (batch, nx, ny), dtype, "sym")
        dt = 5e-5
    opts, integ, n = {}, L.INT_RK4, 6
    if case == "ch_rk4_stage":
        opts["fuse_stages"] = -1
    elif case == "ch_euler_pair":
        integ = L.INT_EULER
    elif case == "ch_euler_odd":
        integ, n = L.INT_EULER, 7
    out = []
    for streams in (1, 2):
        eng = P.HipEngine()
        eng.set_group_envs(group)
        eng.set_group_streams(streams)
        eng.set_graph(-1)
        for k, v in opts.items():
            getattr(eng, "set_" + k)(v)
        eng.configure(dtype=u.dtype, batch=batch, **eq._engine_problem())
        eng.set_state(u)
        eng.advance(integ, dt, n // 2)
        eng.advance(integ, dt, n - n // 2)  # a second call: the second stream starts behind the first call's results
        assert eng.last_group_streams() == streams
        assert eng.last_groups() == -(-batch // group)
        st = eng.reduce(L.RED_MEAN)  # work queued on the ctx stream after the join sees both groups' results
        out.append((eng.get_state(), st))
        eng.close()
    (a, ma), (b, mb) = out
    assert np.isfinite(a).all() and np.any(a != u)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ma, mb)


def test_auto_schedule_runs_large_batches_as_two_side_by_side_halves():
    """auto: a batch that fits the cache as one group runs as two halves side by side once it is large (> 2 M cells);
    small ones keep the single sweep (and the hipGraph replay); results do not depend on it"""
    dom = std_domain(P, 256, 256)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    rng = np.random.default_rng(13)
    u = white_noise_state(rng, (40, 256, 256), np.float32, "sym")
    res = []
    for streams in (0, 1):
        eng = P.HipEngine()
        eng.set_group_streams(streams)
        eng.configure(dtype=u.dtype, batch=40, **eq._engine_problem())
        eng.set_state(u)
        eng.advance(L.INT_RK4, 5e-5, 4)
        res.append((eng.get_state(), eng.last_group_streams(), eng.last_groups()))
        eng.close()
    assert res[0][1:] == (2, 2) and res[1][1:] == (1, 1), (res[0][1:], res[1][1:])
    np.testing.assert_array_equal(res[0][0], res[1][0])
    eng = P.HipEngine()
    eng.configure(dtype=u.dtype, batch=8, **eq._engine_problem())
    eng.set_state(u[:8])
    eng.advance(L.INT_RK4, 5e-5, 4)
    assert eng.last_group_streams() == 1
    eng.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,batch,group", [((64, 64), 5, 2), ((128, 64), 7, 4), ((64, 256), 3, 2)])
def test_imex_grouped_equals_ungrouped_bitwise(dtype, shape, batch, group):
    """IMEX packs two environments per complex field: groups are even-sized, an odd batch leaves the last
    field with a zero imaginary part -- in whichever group it falls."""
    rng = np.random.default_rng(12)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
    u = np.clip(0.5 + 0.01 * rng.standard_normal((batch, nx, ny)), 0.05, 0.95).astype(dtype)
    whole, g1, k1 = _advance(eq, solver, u, L.INT_IMEX, 1e-6, 5, -1)
    parts, g2, k2 = _advance(eq, solver, u, L.INT_IMEX, 1e-6, 5, group)
    assert "imex_fused_lds_fft" in k1 and k1 == k2
    assert g1 == 1 and g2 == -(-batch // ((group + 1) // 2 * 2)), (g1, g2)
    assert np.any(whole != u)
    np.testing.assert_array_equal(parts, whole)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,batch,group", [((64, 64), 5, 2), ((128, 64), 7, 3), ((64, 256), 3, 2)])
def test_strang_grouped_equals_ungrouped_bitwise(dtype, shape, batch, group):
    """per-environment interaction strengths and per-environment potentials ride along with the groups"""
    nx, ny = shape
    dom = P.Domain((nx, ny), ((-12.0, 12.0), (-9.0, 9.0)), "dimensionless")
    eq = P.GPE2DTSControl(dom, 500.0, 0.2, lambda t, x, y: 0.05 * x - 0.02 * y, trap_factor=1.0, kinetic=True)
    X, Y = dom.mesh()
    states = []
    for b in range(batch):
        psi = np.exp(-(X**2 / (14.0 + b) + Y**2 / 10.0)) * np.exp(0.3j * X - 0.1j * b * Y)
        psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * dom.dx[0] ** 2)
        states.append(np.stack([psi.real, psi.imag], axis=-1))
    u = np.stack(states).astype(dtype)
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    outs = {}
    for g in (-1, group):
        eng = P.HipEngine()
        eng.set_group_envs(g)
        eng.configure(dtype=dtype, batch=batch, **eq._engine_problem())
        pots = np.stack([eq.potential(0.0) * (1.0 + 0.05 * b) for b in range(batch)])
        eng.set_aux(L.AUX_GPE_POTENTIAL, pots, per_env=True)
        eng.set_env_gpe_k(0, 500.0 + 25.0 * np.arange(batch))
        solver.configure_engine(eng, eq)
        eng.set_state(u)
        eng.advance(L.INT_STRANG, 1e-3, 4)
        outs[g] = eng.get_state()
        assert eng.last_kernel == "strang_fused_lds_fft"
        assert eng.last_groups() == (1 if g < 0 else -(-batch // group))
        eng.close()
    assert np.any(outs[-1] != u)
    np.testing.assert_array_equal(outs[group], outs[-1])
    # and the per-environment parameters did reach their environments: env b against the oracle
    for b in (0, batch - 1):
        k_b, pot_b = 500.0 + 25.0 * b, eq.potential(0.0) * (1.0 + 0.05 * b)
        bt = lambda t, yy: np.stack([np.zeros_like(pot_b), -(pot_b + k_b * (yy[..., 0] ** 2 + yy[..., 1] ** 2))], axis=-1)
        ref = u[b].astype(np.float64)
        for i in range(4):
            ref = O.strang_step(bt, i * 1e-3, ref, 1e-3, eq.A_term, eq.dx, 1.0)
        assert rel_l2(outs[group][b], ref) < (1e-11 if dtype is np.float64 else 2e-5)


# ------------------------------------------------------------ BASELINE configurations, full batch
REGSOL_C = CO.closure(0, 1, (3.0, -6.0))
C1MC_C = CO.closure(0, 0, (0.0, 1.0, -1.0))


def _ch_ic(n, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95).astype(dtype)


def test_config3_rk4_headline_batch_vs_c_oracle():
    """ch_rk4_1024_f32 exactly as bench.py runs it: 32 environments, auto grouping -> 4 groups of 8, two side by
    side on two streams (256 MiB resident), the whole-substep kernel on 32 x 128 tiles with the XCD-aware block map.  First / last
    environment of each group against oracle/c_oracle.c (fp32 state to 5e-7 absolute, increment to 5e-5 relative)."""
    n, batch, nsub, dt = 1024, 32, 4, 2e-7
    dom = std_domain(P, n, n)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.stack([_ch_ic(n, b) for b in range(batch)])
    out, groups, kernel = _advance(eq, None, y0, L.INT_RK4, dt, nsub, 0)
    assert groups == 4, groups
    assert kernel == "rk4_quad<f32,CH,logit,rows32>", kernel
    hx, hy = dom.dx
    for b in (0, 7, 8, 15, 16, 23, 24, 31):
        ref = CO.rk4(0, y0[b], hx, hy, 0.002, REGSOL_C, C1MC_C, dt, nsub, threads=8)
        assert np.max(np.abs(out[b] - ref)) < 5e-7, (b, np.max(np.abs(out[b] - ref)))
        inc, inc_ref = out[b].astype(np.float64) - y0[b], ref.astype(np.float64) - y0[b]
        assert rel_l2(inc, inc_ref) < 5e-5, (b, rel_l2(inc, inc_ref))
    # environments in between are not copies of their neighbours
    assert np.any(out[1] != out[0]) and np.any(out[17] != out[16])


def test_config3_rk4_f64_batch_vs_c_oracle():
    """fp64 headline variant (16 environments = 4 groups of 4, two side by side)"""
    n, batch, nsub, dt = 1024, 16, 3, 2e-7
    dom = std_domain(P, n, n)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.stack([_ch_ic(n, 100 + b, np.float64) for b in range(batch)])
    out, groups, kernel = _advance(eq, None, y0, L.INT_RK4, dt, nsub, 0)
    assert groups == 4 and "stage_pair<f64,CH" in kernel, (groups, kernel)
    hx, hy = dom.dx
    for b in (0, 3, 4, 7, 8, 11, 12, 15):
        ref = CO.rk4(0, y0[b], hx, hy, 0.002, REGSOL_C, C1MC_C, dt, nsub, threads=8)
        assert rel_l2(out[b] - y0[b], ref - y0[b]) < 1e-10, (b, rel_l2(out[b] - y0[b], ref - y0[b]))


def test_config2_ac_rk4_batch64_vs_c_oracle():
    """BASELINE config 2: Allen-Cahn 512^2 fp32, RK4 dt 5e-5, 64 environments on one GPU (single-pass kernel)."""
    n, batch, nsub, dt = 512, 64, 8, 5e-5
    dom = std_domain(P, n, n)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    y0 = np.stack([(0.01 * np.random.default_rng(b).standard_normal((n, n))).astype(np.float32) for b in range(batch)])
    hx, hy = dom.dx
    cmu, cmob = CO.closure(0, 0, (0.0, -1.0, 0.0, 1.0)), CO.closure(0, 0, (1.0,))
    for group in (0, 24):  # auto (the batch fits the cache: two halves side by side) and a forced split 24 + 24 + 16
        out, groups, kernel = _advance(eq, None, y0, L.INT_RK4, dt, nsub, group)
        assert "rk4_quad" in kernel, kernel
        assert groups == (2 if group == 0 else 3)
        for b in (0, 23, 24, 31, 32, 47, 48, 63):
            ref = CO.rk4(1, y0[b], hx, hy, 0.002, cmu, cmob, dt, nsub, threads=8)
            assert np.max(np.abs(out[b] - ref)) < 5e-8, (b, np.max(np.abs(out[b] - ref)))
            inc, inc_ref = out[b].astype(np.float64) - y0[b], ref.astype(np.float64) - y0[b]
            assert rel_l2(inc, inc_ref) < 5e-5, (b, rel_l2(inc, inc_ref))


def test_config3_imex_batch32_vs_oracle():
    """BASELINE config 3 (ii): CH 1024^2 fp32 IMEX A = 0.5, dt 1e-6, 32 environments per GPU -> 2 groups of 16,
    two environments per complex field.  First / last environment of each group against oracle/np_oracle.py
    (solvers.py:56-63 restated) in fp64 from the same fp32 initial fields."""
    n, batch, nsub, dt = 1024, 32, 3, 1e-6
    dom = std_domain(P, n, n)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
    y0 = np.stack([_ch_ic(n, 200 + b) for b in range(batch)])
    out, groups, kernel = _advance(eq, solver, y0, L.INT_IMEX, dt, nsub, 0)
    assert groups == 2 and "imex_fused_lds_fft" in kernel, (groups, kernel)
    hx, hy = dom.dx
    sym = O.ch_fourier_symbol(n, n, hx, hy, 0.002)
    rhs = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, MU["regsol"], MOB["c1mc"])
    for b in (0, 15, 16, 31):
        ref = y0[b].astype(np.float64)
        for i in range(nsub):
            ref = O.imex_step(rhs, i * dt, ref, dt, 0.5, sym)
        inc, inc_ref = out[b].astype(np.float64) - y0[b], ref - y0[b]
        assert rel_l2(inc, inc_ref) < 5e-5, (b, rel_l2(inc, inc_ref))
        assert np.max(np.abs(out[b] - ref)) < 5e-7


def test_config4_gpe_strang_batch128_vs_oracle():
    """BASELINE config 4: GPE 512^2 complex64, Strang split step, 128 environments on one GPU -> 4 groups of
    32, two side by side.  Every environment starts from its own wave packet; first / last of each group against
    oracle/np_oracle.py (solvers.py:99-115 restated)."""
    n, batch, nsub, dt = 512, 128, 4, 1e-3
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    eq = P.GPE2DTSControl(dom, 1000.0, 0.0, lambda t, x, y: 0.0, trap_factor=1.0, kinetic=True)
    X, Y = dom.mesh()
    y0 = np.empty((batch, n, n, 2), dtype=np.float32)
    for b in range(batch):
        w = 4.0 + 0.01 * b
        psi = np.exp(-((X - 0.02 * b) ** 2 + Y**2) / (2 * w**2)) * np.exp(0.01j * b * X)
        psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * dom.dx[0] ** 2)
        y0[b, ..., 0], y0[b, ..., 1] = psi.real, psi.imag
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    out, groups, kernel = _advance(eq, solver, y0, L.INT_STRANG, dt, nsub, 0)
    assert groups == 4 and kernel == "strang_fused_lds_fft", (groups, kernel)  # 4 groups of 32, two side by side
    bt = lambda t, yy: O.gpe_b_terms(yy, X, Y, 1000.0, 0.0, 1.0, 0.0)
    for b in (0, 31, 32, 63, 64, 95, 96, 127):
        ref = y0[b].astype(np.float64)
        for i in range(nsub):
            ref = O.strang_step(bt, i * dt, ref, dt, eq.A_term, eq.dx, 1.0)
        assert rel_l2(out[b], ref) < 2e-5, (b, rel_l2(out[b], ref))
