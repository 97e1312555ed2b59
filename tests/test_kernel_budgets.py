"""Register / scratch budgets of the occupancy-critical kernels, from the compiler's own report.

csrc/build.py compiles with -Rpass-analysis=kernel-resource-usage and keeps what hipcc reports next to every
object.  A few kernels sit right below a VGPR boundary at which a whole workgroup stops fitting on a CU; crossing
it is silent (the code still runs, 20-30 % slower -- round 2 lost 54 -> 71 us on the 512^2 Strang column pass to a
run-time branch worth four registers).  The numbers below are those boundaries, not the current counts.
No GPU needed: hipcc cross-compiles for gfx950.
"""
import pytest

from pde_opt_amd.csrc import build as B


@pytest.fixture(scope="module")
def res():
    B.build(verbose=False)
    r = B.kernel_resources()
    assert len(r) > 300
    return r


def _pick(res, prefix):
    hit = {k: v for k, v in res.items() if k.startswith(prefix)}
    assert hit, prefix
    return hit


# The one family allowed scratch: the in-kernel adaptive solve at 4 vectors per thread (64 x 128 fp32 / 64 x 64 fp64).
# Six slopes of four vectors are 96 registers before the stencil's own ~100; with the state, k7 and the stage input
# moved to LDS what is left over the 256 of a 512-thread workgroup is <= 42 dwords of rarely touched controller
# state.  The alternatives measured worse on paper: 256 threads x 8 vectors runs one wave per SIMD.
# The multi-workgroup adaptive kernel (stencil_coop_adaptive.hpp; 512 threads, 256 registers each): the fp32 fixed-closure
# instantiations -- the notebook workloads -- hold everything in registers but for one register of step-loop state of the
# smoothed-boundary Cahn-Hilliard form (8 B); the fp64 smoothed-boundary forms spill a
# little (five stage bodies with their hoisted constants in one kernel), the run-time closure walk (Legendre recurrences
# in a loop, fp64) is the rare path and may spill more.
# (last template argument: 0 = the adaptive solve, 1 = the fixed-step mode, which holds no slopes: only its fp64 run-time
# closure walk spills)
SCRATCH_ALLOWED = [("small_tsit5_kernel<", ", 4, 512>", 192), ("tsit5_coop_kernel<float, ", ", true, 0>", 16),
                   ("tsit5_coop_kernel<float, ", ", false, 0>", 96), ("tsit5_coop_kernel<double, ", ", true, 0>", 256),
                   ("tsit5_coop_kernel<double, ", ", false, 0>", 1280), ("tsit5_coop_kernel<double, ", ", false, 1>", 640)]


def test_no_kernel_spills(res):
    spilled = {k: v["scratch"] for k, v in res.items() if v.get("scratch", 0) > 0}
    for k in list(spilled):
        for prefix, suffix, cap in SCRATCH_ALLOWED:
            if k in spilled and k.startswith(prefix) and k.endswith(suffix) and spilled[k] <= cap:
                del spilled[k]
    assert not spilled, spilled


# (kernel name prefix, VGPR budget, why)
BUDGETS = [
    # three 512-thread workgroups (8 waves each, 42-44 KB of LDS) per CU = 6 waves per SIMD
    ("stage_pair_kernel<float, 3, 0, 2, false, 512>", 80, "PAIR_12 fp32: 3 workgroups per CU"),
    ("stage_pair_kernel<float, 3, 1, 2, false, 512>", 80, "PAIR_34 fp32: 3 workgroups per CU"),
    ("stage_pair_kernel<double, 3, 0, 2, false, 512>", 80, "PAIR_12 fp64: 3 workgroups per CU"),
    ("stage_pair_kernel<double, 3, 1, 2, false, 512>", 96, "PAIR_34 fp64: 2 workgroups per CU (3 measured no faster)"),
    ("stage_pair_kernel<float, 3, 2, 2, false, 512>", 64, "PAIR_K (IMEX slope): 8 waves per SIMD"),
    # the Strang / IMEX column pass up to N = 512 runs 1024-thread workgroups, two per CU: 8 waves per SIMD
    ("strang_col_reg_kernel<float, 512, 16, 8, ", 64, "two 1024-thread workgroups per CU"),
    ("strang_col_reg_kernel<float, 256, 16, 8, ", 64, "two 1024-thread workgroups per CU"),
    ("strang_row_reg_kernel<float, 512, ", 64, "8 waves per SIMD"),
    # the 32-point engine at N = 1024: 4 waves per SIMD is what its LDS footprint was halved for
    ("strang_col32_kernel<float, 16, ", 128, "two 512-thread workgroups per CU"),
    ("imex_row_fwd_reg_kernel<float, 1024>", 128, "4 waves per SIMD"),
    ("imex_row_inv_reg_kernel<float, 1024>", 128, "4 waves per SIMD"),
    ("ac_rk4_quad_kernel<4, ", 72, "single-pass Allen-Cahn, 32-row tiles: 7 waves per SIMD"),
    # the headline's kernel: two 1024-thread workgroups per CU (79.5 KB of LDS each) = 8 waves per SIMD
    ("ch_rk4_quad_kernel<", 64, "single-pass Cahn-Hilliard: two 1024-thread workgroups per CU"),
]


@pytest.mark.parametrize("prefix,budget,why", BUDGETS, ids=[b[0] for b in BUDGETS])
def test_vgpr_budget(res, prefix, budget, why):
    for name, v in _pick(res, prefix).items():
        assert v["vgpr"] <= budget, f"{name}: {v['vgpr']} VGPRs > {budget} ({why})"
        assert v["agpr"] == 0
