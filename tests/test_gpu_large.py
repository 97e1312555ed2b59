"""Maximum sizes: batches whose fields hold more than 2^31 ELEMENTS (and so more than 2^33 bytes), where any
32-bit environment / row offset in a kernel, a launch wrapper or a copy would wrap.

The MI355X has 288 GB: a per-GPU batch of thousands of environments is what the memory is sized for (DESIGN
section 3).  Every environment is one of four seeded patterns, uploaded in chunks (no multi-GB host array); the
first and the last environments -- the ones beyond the 32-bit boundary -- are fetched alone and compared with
the CPU oracle, and with each other across the boundary (same pattern => same bits).
"""
import numpy as np
import pytest

import pde_opt_amd as P
from oracle import np_oracle as O
from pde_opt_amd import _lib as L
from util import MOB, MU, inc_tol_f32, rel_l2, std_domain

pytestmark = pytest.mark.gpu

PATTERNS = 4


def _fill(eng, patterns, batch, chunk=64):
    reps = -(-chunk // PATTERNS)
    block = np.concatenate([patterns] * reps)[:chunk]  # chunk is a multiple of PATTERNS
    for lo in range(0, batch, chunk):
        n = min(chunk, batch - lo)
        eng.set_state(block[:n], env_first=lo)


def _ends(eng, batch):
    """environments 0..3 and the last four, (8,) + state_shape"""
    return np.concatenate([eng.get_state(0, PATTERNS), eng.get_state(batch - PATTERNS, PATTERNS)])


@pytest.mark.parametrize("case", ["ch_rk4_pair", "ac_rk4_quad", "ch_euler_generic", "ch_imex"])
def test_fields_beyond_2_31_elements(case):
    n, batch = 1024, 2052  # 2052 x 2^20 cells = 2.15e9 > 2^31 elements per field; 8.6 GB per fp32 field
    assert batch % PATTERNS == 0 and batch * n * n > 2**31
    rng = np.random.default_rng(77)
    dom = std_domain(P, n, n)
    hx, hy = dom.dx
    solver, opts = None, {}
    if case.startswith("ch"):
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        pats = np.clip(0.5 + 0.05 * rng.standard_normal((PATTERNS, n, n)), 0.05, 0.95).astype(np.float32)
        f = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, MU["regsol"], MOB["c1mc"])
        dt = 2e-7
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
        pats = (0.1 * rng.standard_normal((PATTERNS, n, n))).astype(np.float32)
        f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
        dt = 5e-5
    integ, step, nsub = L.INT_RK4, O.rk4_step, 2
    if case == "ch_euler_generic":
        integ, step, opts = L.INT_EULER, O.euler_step, {"kernel_path": L.PATH_GENERIC}
    if case == "ch_imex":
        solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft)
        sym = O.ch_fourier_symbol(n, n, hx, hy, 0.002)
        integ, dt = L.INT_IMEX, 1e-6
        step = lambda ff, t, y, h: O.imex_step(ff, t, y, h, 0.5, sym)
    eng = P.HipEngine()
    for k, v in opts.items():
        getattr(eng, "set_" + k)(v)
    eng.configure(dtype=np.float32, batch=batch, **eq._engine_problem())
    eq._engine_upload(eng, 0.0)
    if solver is not None:
        solver.configure_engine(eng, eq)
    _fill(eng, pats, batch)
    np.testing.assert_array_equal(_ends(eng, batch), np.concatenate([pats, pats]))  # the copies themselves
    eng.advance(integ, dt, nsub)
    got = _ends(eng, batch)
    kernel, groups = eng.last_kernel, eng.last_groups()
    nonfinite = float(eng.reduce(L.RED_NONFINITE).sum())
    eng.close()
    assert nonfinite == 0.0
    assert groups > 1, (kernel, groups)  # the cache-resident group loop ran, with offsets beyond 2^31
    if case == "ch_rk4_pair":
        assert "stage_pair" in kernel or "rk4_quad" in kernel, kernel
    if case == "ac_rk4_quad":
        assert "rk4_quad" in kernel, kernel
    if case == "ch_imex":
        assert "imex_fused_lds_fft" in kernel, kernel
    np.testing.assert_array_equal(got[PATTERNS:], got[:PATTERNS])  # same pattern on both sides of the boundary
    for b in (0, PATTERNS - 1):
        ref = pats[b].astype(np.float64)
        for i in range(nsub):
            ref = step(f, i * dt, ref, dt)
        last = got[PATTERNS + b].astype(np.float64)
        assert np.max(np.abs(last - ref)) < 1e-6, (case, b, np.max(np.abs(last - ref)))
        assert rel_l2(last - pats[b], ref - pats[b]) < inc_tol_f32(ref, pats[b]), (case, b, rel_l2(last - pats[b], ref - pats[b]))


def test_strang_beyond_2_31_elements():
    """GPE 512^2 c64: 4104 environments x 2^19 real elements = 2.15e9; per-environment interaction strengths ride
    along, the last environment differs from the first only through its own k"""
    n, batch = 512, 4104
    assert batch * n * n * 2 > 2**31
    dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    lights = lambda t, x, y: 0.05 * x - 0.02 * y
    eq = P.GPE2DTSControl(dom, 500.0, 0.2, lights, trap_factor=1.0, kinetic=True)
    X, Y = dom.mesh()
    pats = []
    for b in range(PATTERNS):
        psi = np.exp(-(X**2 / (14.0 + b) + Y**2 / 10.0)) * np.exp(0.3j * X - 0.1j * b * Y)
        psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * dom.dx[0] ** 2)
        pats.append(np.stack([psi.real, psi.imag], axis=-1))
    pats = np.stack(pats).astype(np.float32)
    ks = np.full(batch, 500.0)
    ks[-1] = 650.0
    solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
    eng = P.HipEngine()
    eng.configure(dtype=np.float32, batch=batch, **eq._engine_problem())
    eq._engine_upload(eng, 0.0)
    eng.set_env_gpe_k(0, ks)
    solver.configure_engine(eng, eq)
    _fill(eng, pats, batch)
    eng.advance(L.INT_STRANG, 1e-3, 2)
    got = _ends(eng, batch)
    assert eng.last_kernel == "strang_fused_lds_fft" and eng.last_groups() > 1
    eng.close()
    np.testing.assert_array_equal(got[PATTERNS:2 * PATTERNS - 1], got[:PATTERNS - 1])
    A_term = np.asarray(eq.A_term)
    for b, k in ((0, 500.0), (2 * PATTERNS - 1, 650.0)):
        bt = lambda t, y: O.gpe_b_terms(y, X, Y, k, 0.2, 1.0, lights(t, X, Y))
        ref = pats[b % PATTERNS].astype(np.float64)
        for i in range(2):
            ref = O.strang_step(bt, i * 1e-3, ref, 1e-3, A_term, dom.dx[0], 1.0)
        assert rel_l2(got[b], ref) < 2e-5, (b, rel_l2(got[b], ref))
