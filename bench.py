"""bench.py -- env-steps/sec and achieved HBM GB/s of the fused Cahn-Hilliard RK4 step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): env-steps/sec & achieved HBM GB/s, Cahn-Hilliard 1024^2 RK4.
Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  1024^2 fp32 field, kappa=0.002, mu = log(c/(1-c)) + 3(1-2c), D = c(1-c), IC
  clip(0.5 + 0.01 N(0,1), 0.05, 0.95) seeded per environment, explicit RK4 dt=2e-7,
  100 substeps per environment step, 32 environments per GPU (256 over 8 GPUs).
A "step" = one environment step of every environment on the rank = 100 RK4 substeps.
Weak scaling: per-GPU work is fixed; environments are independent, there is no data-path
collective; torch.distributed (RCCL) only carries the barrier and the max-over-ranks.

Inputs are resident in HBM when the timed region starts (states uploaded before warm-up).
Timing: barrier + device sync on both sides, max over ranks.  The roofline figure is measured live
with HIP events recorded on the engine's own stream (pdeopt_timer_start/stop) and the number of
kernel launches counted by the library (pdeopt_get_counter).

The `roofline` object says what binds the dominant kernel (roofline_block below).  SURVEY 8(d)'s ALGORITHMIC
byte count (RK4 with one fused kernel per stage: 16 words/cell/substep = 64 B fp32) over the HIP-event launch time
is always reported (`algorithmic_gbs`); the shipped kernels fuse stage PAIRS (7 words/cell/substep of real
traffic) and keep environment groups resident in the 256 MiB Infinity Cache, so that figure exceeds the HBM peak
-- removed traffic, not a measurement error -- and the headline kernel is bound by VALU issue instead:
`bound = "valu"`, achieved / peak in wave64 VALU instructions per second (SQ_INSTS_VALU per launch from the
committed PMC profile of this command, profiles/pmc_r02.json, over the live launch time; peak = the scalar-fp32
issue rate measured by tools/valubench.hip, profiles/r02_valubench.txt), `traffic` = measured L2 fabric-side bytes.

After the timed region the line proves its own result (`parity_spot_*`: the same library call on fresh inputs,
first / last environment of every group against the CPU oracle) and adds `api_value`: env-steps/s through
VectorPDEEnv.step with the reward and uint8 observations formed on the GPU.

Other workloads (--workload) are secondary rows for DESIGN.md, not the bench line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

# algorithmic words per cell per substep; a word is one real scalar
WORDS = {"rk4": 16, "euler": 2, "imex": 9, "strang": 22,  # strang: 88 B / 4 B at c64
         "rk4_sbm": 28}  # smoothed boundary: + psi, |grad psi|/psi and the mask per stage ("+8 B/cell/stage" + mask)

METRIC = "env-steps/sec (Cahn-Hilliard 1024^2 RK4, 100 substeps/env-step) & achieved HBM GB/s"
WORKLOADS = {
    "ch_rk4_1024_f32": dict(eq="ch", n=1024, dtype=np.float32, integ="rk4", dt=2e-7, substeps=100, batch=32),
    # same kernel with the double-well closures (mu = c^3 - c, D = 1 + c^2: no log / rcp in mu)
    # (D <= 2 here against <= 0.25 for c (1 - c): the explicit stability limit of the biharmonic is 8 x tighter, dt 2e-7 diverges)
    "ch_rk4_1024_f32_cubic": dict(eq="ch", n=1024, dtype=np.float32, integ="rk4", dt=5e-8, substeps=100, batch=32,
                                  closures="cubic"),
    "ch_rk4_1024_f64": dict(eq="ch", n=1024, dtype=np.float64, integ="rk4", dt=2e-7, substeps=100, batch=16),
    # the reference's own grid sizes (tests/test_solvers.py:25,68,145; notebooks: 32^2 ... 128^2): the whole-environment-
    # step kernel (csrc/stencil_small.hpp), one launch per environment step, one compute unit per environment
    "ch_rk4_64_f32_small": dict(eq="ch", n=64, dtype=np.float32, integ="rk4", dt=2e-7, substeps=100, batch=256),
    "ch_rk4_128_f32_small": dict(eq="ch", n=128, dtype=np.float32, integ="rk4", dt=2e-7, substeps=100, batch=256),
    "ac_rk4_64_f32_small": dict(eq="ac", n=64, dtype=np.float32, integ="rk4", dt=5e-5, substeps=100, batch=256),
    # ONE mid-sized environment (single-environment latency: the RL loop of pde_env.py:244-317 on one 96^2 / 128^2 field):
    # several compute units per environment, 1-2 substeps per neighbour exchange (csrc/stencil_coop_adaptive.hpp, MODE 1)
    "ch_rk4_96_f32_1env": dict(eq="ch", n=96, dtype=np.float32, integ="rk4", dt=2e-7, substeps=100, batch=1),
    "ch_rk4_128_f32_1env": dict(eq="ch", n=128, dtype=np.float32, integ="rk4", dt=2e-7, substeps=100, batch=1),
    # smoothed-boundary Cahn-Hilliard (cahn_hilliard.py:204-289) on the LDS-tiled kernel: a disc-shaped
    # level set, regular-solution free energy, contact angle and boundary flux varying in time (notebooks/smooth_boundary.ipynb)
    "ch_sbm_1024_f32": dict(eq="ch_sbm", n=1024, dtype=np.float32, integ="rk4", dt=2e-3, substeps=100, batch=8, words="rk4_sbm",
                            abs_tol=2e-6),  # dt = 2e-3: the state moves 10x as far per substep as in the periodic workloads
    # Cahn-Hilliard in three dimensions (cahn_hilliard.py:113-200): the two-pass kernels (mu field, then
    # the flux divergence + Runge-Kutta update), 8 environments of 128^3
    "ch3d_rk4_128_f32": dict(eq="ch3d", n=128, dtype=np.float32, integ="rk4", dt=2e-7, substeps=20, batch=8),
    "ac_rk4_512_f32": dict(eq="ac", n=512, dtype=np.float32, integ="rk4", dt=5e-5, substeps=100, batch=64),
    "ch_imex_1024_f32": dict(eq="ch", n=1024, dtype=np.float32, integ="imex", dt=1e-6, substeps=100, batch=32),
    # (abs_tol: the spot check's bound on the largest absolute state error; the wavefunction is O(0.2) and its relative
    # error -- not an increment's -- is what the 2e-5 gate holds: 2.6e-6 observed = 5.1e-7 absolute)
    "gpe_strang_512_c64": dict(eq="gpe", n=512, dtype=np.float32, integ="strang", dt=1e-3, substeps=100, batch=128, abs_tol=1e-6),
    # the same with a time-dependent control: every environment's lights(t, x, y) is a moving Gaussian spot
    # evaluated in-kernel at each substep's t0 (pdeopt_set_gpe_spots)
    "gpe_strang_512_c64_spots": dict(eq="gpe", n=512, dtype=np.float32, integ="strang", dt=1e-3, substeps=100, batch=128,
                                     spots=True, abs_tol=1e-6),
}

REGSOL = lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c)  # noqa: E731
C1MC = lambda c: c * (1 - c)  # noqa: E731
# smoothed-boundary workload: free energy, contact-angle ramp and boundary flux (the closures of tests/util.py)
SBM_F = lambda c: c * np.log(c) + (1.0 - c) * np.log(1.0 - c) + 3.0 * c * (1.0 - c) + 0.059  # noqa: E731
SBM_THETA = lambda t: 34.9065850398866 * t**2 - 10.4719755119660 * t + np.pi / 2  # noqa: E731
SBM_FLUX = lambda t: 0.02 * (1.0 + 3.0 * t)  # noqa: E731


def sbm_psi(nx, ny, floor=0.05):
    """disc-shaped level set in (floor, 1] on a unit-spacing grid"""
    x, y = np.arange(nx) + 0.5, np.arange(ny) + 0.5
    X, Y = np.meshgrid(x, y, indexing="ij")
    r = np.sqrt((X - 0.5 * nx) ** 2 + (Y - 0.5 * ny) ** 2)
    return floor + (1.0 - floor) * 0.5 * (1.0 + np.tanh((0.3 * min(nx, ny) - r) / 2.5))


def make_problem(P, name, batch, rank):
    w = WORKLOADS[name]
    n, dtype = w["n"], w["dtype"]
    if w["eq"] == "gpe":
        dom = P.Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
        lights = P.GaussianSpots.moving(20.0, (-3.0, 0.0), (3.0, 1.0), 0.1, 1.5) if w.get("spots") else (lambda t, x, y: 0.0)
        eq = P.GPE2DTSControl(dom, 1000.0, 0.0, lights, trap_factor=1.0, kinetic=True)
        X, Y = dom.mesh()
        y0 = np.empty((batch, n, n, 2), dtype=dtype)
        for b in range(batch):  # normalised Gaussians, width L/6, a little different per environment
            wdt = 4.0 * (1.0 + 0.002 * (rank * batch + b))
            psi = np.exp(-(X**2 + Y**2) / (2 * wdt**2))
            psi /= np.sqrt(np.sum(psi**2) * dom.dx[0] ** 2)
            y0[b, ..., 0], y0[b, ..., 1] = psi, 0.0
        solver = P.StrangSplitting(eq.A_term, eq.dx, eq.fft, eq.ifft, 1.0)
        return eq, y0, solver
    if w["eq"] == "ch_sbm":
        import types

        psi = sbm_psi(n, n)
        dom = P.Domain((n, n), ((0.0, float(n)), (0.0, float(n))), "dimensionless", geometry=types.SimpleNamespace(smooth=psi))
        eq = P.CahnHilliard2DSmoothedBoundary(dom, 1.5, SBM_F, REGSOL, C1MC, SBM_THETA, SBM_FLUX)
        y0 = np.empty((batch, n, n), dtype=dtype)
        for b in range(batch):
            y0[b] = np.clip(0.5 + 0.1 * np.random.default_rng(rank * batch + b).standard_normal((n, n)), 0.1, 0.9)
        return eq, y0, P.RK4()
    L_ = 0.01 * n
    if w["eq"] == "ch3d":
        dom = P.Domain((n, n, n), ((-L_ / 2, L_ / 2),) * 3, "dimensionless")
        y0 = np.empty((batch, n, n, n), dtype=dtype)
        for b in range(batch):
            y0[b] = np.clip(0.5 + 0.01 * np.random.default_rng(rank * batch + b).standard_normal((n, n, n)), 0.05, 0.95)
        return P.CahnHilliard3DPeriodic(dom, 0.002, REGSOL, C1MC), y0, P.RK4()
    dom = P.Domain((n, n), ((-L_ / 2, L_ / 2), (-L_ / 2, L_ / 2)), "dimensionless")
    y0 = np.empty((batch, n, n), dtype=dtype)
    for b in range(batch):
        rng = np.random.default_rng(rank * batch + b)  # seeds 0..255 over 8 GPUs x 32 envs
        if w["eq"] == "ch":
            y0[b] = np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95)
        else:
            y0[b] = 0.01 * rng.standard_normal((n, n))
    if w["eq"] == "ch" and w.get("closures") == "cubic":
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: 1 + c**2)
    elif w["eq"] == "ch":
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, REGSOL, C1MC)
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
    solver = P.SemiImplicitFourierSpectral(0.5, eq.fourier_symbol, eq.fft, eq.ifft) if w["integ"] == "imex" else P.RK4()
    return eq, y0, solver


# Parity gates of the spot checks (relative L2 of the state INCREMENT against the fp64 oracle; observed in round 2:
# RK4 6.4e-6, IMEX 4.1e-6, Strang 2.6e-6) and the hard bound on the largest absolute state error of an fp32 run
# (observed 3.6e-7 on states of O(1): a few fp32 roundings of the state per 100 substeps)
SPOT_TOL_F32 = {"rk4": 5e-5, "imex": 5e-5, "strang": 2e-5}
SPOT_TOL_F64 = {"rk4": 1e-9, "imex": 1e-8, "strang": 1e-10}
SPOT_ABS_TOL_F32 = 5e-7

# per-launch PMC averages of this round's build (tools/pmc_to_json.py); the previous round's while none is committed yet
PMC_FILE = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("pmc_r04.json", "pmc_r03.json", "pmc_r02.json")) if os.path.exists(f)),
                os.path.join(ROOT, "profiles", "pmc_r04.json"))
N_SIMD, SHADER_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md chip table
# What the VALU can do, measured IN the kernel (tools/valubench.hip: s_memtime / s_memrealtime stamps around >= 1 ms bodies,
# profiles/r03_valubench_raw.txt): a SIMD issues one wave64 fp32 VALU instruction per ~2 shader cycles once >= 3 waves are
# resident (a lone wave: 5.5), v_log / v_rcp per ~4-6, v_pk_fma_f32 per ~2.2 per FMA -- as MI355X_MICROARCH.md says --
# while the chip holds 1.9-2.0 GHz under an FMA stream, not 2.4 (round 2 divided wall times by 2.4 GHz and read 3.5 "clk").
VALU_CLK_MEASURED = 2.0      # shader cycles per wave64 scalar-fp32 VALU instruction per SIMD, >= 3 resident waves
VALU_CLK_MEASURED_F64 = 2.5  # v_fma_f64 at 8 waves per SIMD (3.4 at 4)
# ... and what the counters read for it (tools/valu_pmc_calib.sh, profiles/r03_valu_pmc_calibration.txt): SQ_ACTIVE_INST_VALU
# books exactly 4 cycles per instruction, so "4 x SQ_ACTIVE_INST_VALU / SIMDs over GRBM_GUI_ACTIVE / 8" reads 1.40 (2-4
# waves per SIMD) to 1.62-1.65 (8 waves) for a kernel that issues NOTHING but independent v_fma_f32 / v_add_f32 / v_mov --
# not 1.0.  The VALU utilisation of a kernel is its reading over that saturated reading.
VALU_BUSY_SATURATED = 1.62
VALU_BUSY_SATURATED_F64 = 0.94  # v_fma_f64 / v_add_f64 take 4.2-4.3 cycles per instruction: an all-fp64 kernel reads 0.92-0.95


def roofline_block(workload, kernel_name, bytes_per_launch, words, avg_launch_s, launches, concurrent=1, shader_hz=None):
    """HBM roofline by SURVEY 8(d)'s algorithmic bytes and, where a PMC profile of this workload is committed, the
    VALU-issue roofline of the same kernel; `bound` names the larger fraction (what binds the kernel).
    `concurrent`: launches in flight side by side (two environment groups on two streams, PDEOPT_OPT_GROUP_STREAMS):
    avg_launch_s is then ONE launch's duration -- what a kernel trace shows -- and the chip moves `concurrent` launches'
    bytes in that time."""
    alg_gbs = concurrent * bytes_per_launch / avg_launch_s / 1e9
    r = {
        "bound": "hbm",
        "kernel": kernel_name + " (average over the launches of a substep)",
        "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
        "traffic": None,
        "algorithmic_gbs": alg_gbs,
        "algorithmic_frac_of_hbm_peak": alg_gbs / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": bytes_per_launch,
        "algorithmic_words_per_cell_substep": words,
        "avg_launch_us": avg_launch_s * 1e6,
        "launches_timed": launches,
        "concurrent_launches": concurrent,
    }
    try:
        pmc = json.load(open(PMC_FILE)).get(workload)
    except Exception:
        pmc = None
    if not pmc:
        r["note"] = "achieved = SURVEY 8(d) algorithmic bytes / HIP-event time of the average launch; no PMC profile of this workload is committed"
        return r
    c = pmc["counters_per_launch"]
    if pmc.get("hbm_bytes_per_launch"):
        r["traffic"] = pmc["hbm_bytes_per_launch"]
        r["traffic_gbs"] = concurrent * r["traffic"] / avg_launch_s / 1e9
        r["traffic_frac"] = r["traffic_gbs"] / HBM_PEAK_GBS
    # cycles are priced on the clock the chip HELD over the timed region (pdeopt_timer_clock: s_memtime over s_memrealtime
    # stamps beside the timer events), not on a data-sheet figure; the 2.4 GHz fallback only when the probe gave nothing
    hz = shader_hz if shader_hz and shader_hz > 1e8 else SHADER_HZ
    r["shader_clock_hz"] = hz
    r["shader_clock_source"] = "measured in this run (s_memtime / s_memrealtime over the timed region)" if hz is shader_hz else "fallback 2.4 GHz (probe unavailable)"
    if c.get("SQ_INSTS_VALU"):
        per_simd = concurrent * c["SQ_INSTS_VALU"] / N_SIMD
        cycles = avg_launch_s * hz
        valu_clk = VALU_CLK_MEASURED_F64 if WORKLOADS.get(workload, {}).get("dtype") is np.float64 else VALU_CLK_MEASURED
        r["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
        r["valu_clk_per_inst_measured"] = valu_clk
        r["frac_valu_measured_issue"] = per_simd * valu_clk / cycles
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
            # quad-cycles summed over SIMDs vs the launch's cycles under the profiler (sum over 8 XCDs)
            r["valu_busy_frac_pmc"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / N_SIMD / (c["GRBM_GUI_ACTIVE"] / 8.0)
            sat = VALU_BUSY_SATURATED_F64 if WORKLOADS.get(workload, {}).get("dtype") is np.float64 else VALU_BUSY_SATURATED
            r["valu_busy_saturated_pmc"] = sat
            # rocprofv3 --pmc serialises launches: the counters' own busy share is that of a launch running ALONE
            r["valu_util_pmc_solo_launch"] = r["valu_busy_frac_pmc"] / sat
            # the run that is timed keeps `concurrent` launches in flight: the same busy-cycle count (4 per instruction)
            # over the LIVE cycles a launch's share of the region lasted, at the clock held live
            r["valu_util"] = concurrent * 4.0 * c["SQ_ACTIVE_INST_VALU"] / N_SIMD / cycles / sat
        # which pipe is busier: the VALU (its counter reading over the reading of an all-VALU kernel) or the fabric (measured
        # traffic over the HBM peak)?
        valu_share = r.get("valu_util", r["frac_valu_measured_issue"])
        if valu_share > (r.get("traffic_frac") or 0.0):
            r["bound"] = "valu"
            r["unit"] = "VALU utilisation (SQ_ACTIVE_INST_VALU share of SIMD cycles / what an all-VALU kernel reads: 1.62 fp32, 0.94 fp64)"
            r["achieved"] = valu_share
            r["peak"] = 1.0
            r["frac"] = valu_share
        r["valu_ginst_per_s"] = concurrent * c["SQ_INSTS_VALU"] / avg_launch_s / 1e9
        r["valu_ginst_per_s_at_measured_issue_cost"] = N_SIMD * hz / valu_clk / 1e9
    if r["bound"] == "hbm" and r.get("traffic_gbs"):
        # memory-bound with measured traffic: price the bytes that moved, not the per-stage byte count (which fusion
        # undercuts -- a fraction above 1 of a hardware peak would say nothing)
        r["achieved"], r["frac"] = r["traffic_gbs"], r["traffic_frac"]
    r["pmc_source"] = os.path.relpath(PMC_FILE, ROOT) + ": " + pmc.get("source", "")
    r["note"] = ("bound = the busier of two pipes: fabric traffic (measured bytes per launch / live launch time, against the 8 TB/s HBM peak; "
                 "Infinity-Cache hits included) and the VALU.  valu_util = VALU-busy cycles (4 x SQ_ACTIVE_INST_VALU per SIMD, from the COMMITTED "
                 "rocprofv3 --pmc profile of this command, " + os.path.relpath(PMC_FILE, ROOT) + ": counters need rocprofv3, the driver's run has none) x the "
                 "launches in flight, over the LIVE launch time x the shader clock measured in this run, over what an all-VALU kernel reads "
                 "(1.62 fp32 / 0.94 fp64, profiles/r03_valu_pmc_calibration.txt).  valu_util_pmc_solo_launch is the counters' own busy share: "
                 "rocprofv3 serialises the launches, so it describes a launch running alone -- a PMC pass with two launches in flight cannot be "
                 "taken.  frac_valu_measured_issue = SQ_INSTS_VALU x the in-kernel-measured cycles per instruction (tools/valubench.hip) over the "
                 "same live cycles.  algorithmic_gbs is SURVEY 8(d)'s per-stage byte count (16 words per cell and RK4 substep) over the same time: "
                 "it exceeds the HBM peak where stages are fused in LDS -- the whole-substep kernels move 2 words per cell and substep "
                 "(traffic), the stage pairs 7.")
    return r


def api_throughput(P, name, rank, steps, warmup):
    """env-steps/s through the Python API the RL loop calls: VectorPDEEnv.step (per-environment control update,
    equation rebuild, one pdeopt_advance) with the reward reduced and the uint8 observation frames quantised on
    the GPU -- the same workload, timed on the wall clock after the raw measurement."""
    w = WORKLOADS[name]
    if w["eq"] not in ("ch", "ac") or w["integ"] not in ("rk4", "imex"):
        return None
    n, batch = w["n"], w["batch"]
    L_ = 0.01 * n
    dom = P.Domain((n, n), ((-L_ / 2, L_ / 2), (-L_ / 2, L_ / 2)), "dimensionless")

    def reset(domain, seed=0):
        rng = np.random.default_rng(seed)
        if w["eq"] == "ch":
            return np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95).astype(w["dtype"])
        return (0.01 * rng.standard_normal((n, n))).astype(w["dtype"])

    if w["eq"] == "ch":
        cubic = w.get("closures") == "cubic"
        eq_t, static = P.CahnHilliard2DPeriodic, {"mu": (lambda c: c**3 - c) if cubic else REGSOL,
                                                   "D": (lambda c: 1 + c**2) if cubic else C1MC}
    else:
        eq_t, static = P.AllenCahn2DPeriodic, {"mu": lambda c: c**3 - c, "R": lambda c: np.ones_like(c)}
    imex = w["integ"] == "imex"
    actions = [1] * batch if imex else [(b % 3) for b in range(batch)]  # IMEX shares one implicit operator

    def run(**obs_kw):
        env = P.VectorPDEEnv(
            batch, eq_t, dom, P.SemiImplicitFourierSpectral if imex else P.RK4, end_time=1e9,
            step_dt=w["dt"] * w["substeps"], numeric_dt=w["dt"], state_to_observation_func=lambda s_: s_,
            reward_function=lambda s_: 0.0, reset_func=reset, reset_control_value=0.002,
            update_control_value=lambda off, old: old + off, update_control_parameter=lambda old, new: new,
            action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -1e-5, 1: 0.0, 2: 1e-5}},
            static_equation_parameters=static, control_equation_parameter_name="kappa",
            solver_parameters={"A": 0.5} if imex else {}, device=int(os.environ.get("LOCAL_RANK", "0")),
            device_reward="var", device_observation=(0.0, 1.0), **obs_kw)
        env.reset(seed=rank * batch)
        for _ in range(warmup):
            env.step(actions)
        t0 = time.perf_counter()
        for _ in range(steps):
            obs, rew, *_ = env.step(actions)
        el = time.perf_counter() - t0
        ok = bool(np.isfinite(rew).all()) and tuple(obs.shape) == (batch, 1, n, n) and "uint8" in str(obs.dtype)
        env.close()
        return el, ok

    el, ok = run(reuse_observation_buffer=True)
    out = {"api_value": batch * steps / el, "api_ms_per_step": 1e3 * el / steps, "api_ok": ok,
           "api": "VectorPDEEnv.step, per-environment kappa control, device variance reward + uint8 frames "
                  f"(1 byte/cell D2H into one reused page-locked buffer), {steps} steps after {warmup} warm-up, one GPU"}
    try:  # frames handed to a consumer on the same GPU as a zero-copy torch tensor: nothing crosses PCIe
        el_d, ok_d = run(observations_on_device=True)
        out.update({"api_value_device_obs": batch * steps / el_d, "api_device_obs_ok": ok_d})
    except ImportError:
        pass
    return out


def decomp_roofline(bytes_per_gpu, elapsed, launches, tile_shape, kernel_name, concurrent=1):
    """The decomposed field's roofline: SURVEY 8(d)'s algorithmic bytes of one GPU's share over the WALL time of the
    substep loop (halo exchange included), and -- where a PMC profile OF THIS WORKLOAD AT THIS TILE SIZE is committed
    (profiles/pmc_<round>.json, key ch_rk4_decomp_tile<nx>x<ny>: rocprofv3 --pmc passes of `bench.py --workload
    ch_rk4_4096_decomp --decomp-grid <nx>`) -- the fabric traffic and VALU-busy share of the kernel(s) that ran, per launch,
    over the wall time per launch.  No other workload's counters are scaled in (round 3 did that and priced stage-pair
    launches with the whole-substep kernel's counters).  `launches`: stencil launches of ONE rank in the timed region;
    `concurrent`: ranks sharing the GPU (virtual ranks), whose launches run side by side."""
    alg_gbs = bytes_per_gpu / elapsed / 1e9
    launch_s = elapsed / max(launches, 1)
    key = "ch_rk4_decomp_tile%dx%d" % tuple(tile_shape)
    r = {"bound": "hbm", "kernel": kernel_name, "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
         "traffic": None, "algorithmic_gbs": alg_gbs, "algorithmic_frac_of_hbm_peak": alg_gbs / HBM_PEAK_GBS,
         "launches_timed": launches, "concurrent_launches": concurrent, "avg_launch_us": launch_s * 1e6,
         "note": "per-GPU share of SURVEY 8(d)'s algorithmic bytes over the WALL time of the substep loop (exchange included); "
                 "avg_launch_us = wall time per stencil launch of one rank; no PMC profile of this workload at this tile size (" + key + ") is committed"}
    try:
        pmc = json.load(open(PMC_FILE)).get(key)
    except Exception:
        pmc = None
    if pmc and pmc.get("hbm_bytes_per_launch"):
        c = pmc["counters_per_launch"]
        r["traffic"] = pmc["hbm_bytes_per_launch"]
        r["traffic_gbs"] = concurrent * r["traffic"] / launch_s / 1e9
        r["traffic_frac"] = r["traffic_gbs"] / HBM_PEAK_GBS
        r["achieved"], r["frac"] = r["traffic_gbs"], r["traffic_frac"]
        r["pmc_kernels"] = pmc.get("kernels")
        if c.get("SQ_ACTIVE_INST_VALU"):
            # VALU-busy cycles of the launch (4 per instruction) over the wall cycles per launch, against an all-VALU kernel's reading
            r["valu_util"] = concurrent * 4.0 * c["SQ_ACTIVE_INST_VALU"] / N_SIMD / (launch_s * SHADER_HZ) / VALU_BUSY_SATURATED
            r["valu_insts_per_launch"] = c.get("SQ_INSTS_VALU")
            if r["valu_util"] > r["traffic_frac"]:
                r.update(bound="valu", achieved=r["valu_util"], peak=1.0, frac=r["valu_util"],
                         unit="VALU utilisation (SQ_ACTIVE_INST_VALU share of SIMD cycles / what an all-VALU kernel reads: 1.62)")
        r["pmc_source"] = os.path.relpath(PMC_FILE, ROOT) + ": " + pmc.get("source", "")
        r["note"] = ("counters of THIS workload's own kernel(s) at this tile size (" + key + ", per launch, committed rocprofv3 --pmc profile) "
                     "over the live WALL time per launch of the substep loop, halo exchange included; algorithmic_gbs = SURVEY 8(d)'s 64 B "
                     "per cell and substep over the same time")
    return r


DECOMP_GRIDS = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}


def run_decomp(args, P, world, rank, local_rank, dist, make_solver=None):
    """--workload ch_rk4_4096_decomp (BASELINE config 5): ONE Cahn-Hilliard field of 4096^2 cells, RK4 dt 2e-7,
    decomposed over px x py ranks (1x1, 2x1, 2x2, 4x2 for N = 1, 2, 4, 8).  Default driver: the library's own substep
    loop on its own RCCL communicator, halo-8 layout -- ONE all-gather of packed halo strips per substep, the strip
    written by the second stage pair's edge tiles (pde_opt_amd/decomp.py, csrc/comm.hip).  --virtual-ranks R runs R
    ranks of ONE process on ONE GPU (in-process group: the same loop, device copies instead of RCCL): every N > 1
    code path without a second GPU.  A step = 100 substeps of the one field; strong scaling (the field is fixed).
    ``make_solver(eq, grid)``: test hook (tests/test_dist_cpu.py drives this function's control flow -- which rank
    runs which collective -- under gloo with an oracle-backed tile)."""
    from pde_opt_amd.decomp import (CartesianGrid, DecomposedSolver, LocalGroupComm, NativeComm, PeerMappedComm, TorchComm,
                                    advance_group)

    n, dt, substeps = args.decomp_grid, 2e-7, args.decomp_substeps
    on_gpu = dist is None or dist.get_backend() == "nccl"
    vranks = args.virtual_ranks
    if vranks and world > 1:
        raise SystemExit("--virtual-ranks runs in ONE process on one GPU")
    nranks = vranks or world
    if nranks not in DECOMP_GRIDS:
        raise SystemExit(f"ch_rk4_4096_decomp runs on {sorted(DECOMP_GRIDS)} ranks, not {nranks}")
    px, py = DECOMP_GRIDS[nranks]
    dom = P.Domain((n, n), ((-0.005 * n, 0.005 * n),) * 2, "dimensionless")
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, REGSOL, C1MC)
    # under torchrun: the library's own RCCL communicator and in-library substep loop (auto / native); the torch
    # all-gather drivers stay selectable for comparison.  The overlap / graph drivers are halo-4 (one exchange per
    # stage pair); everything else takes the halo-8 layout (one per substep).
    native = args.decomp_mode in ("auto", "native", "native-overlap")
    halo = args.decomp_halo or (4 if args.decomp_mode in ("native-overlap", "overlap", "graph") else None)
    rng = np.random.default_rng(0)  # every rank draws the same global field and keeps its tile
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95).astype(np.float32)
    if make_solver is not None:
        sols = [make_solver(eq, CartesianGrid(px, py, rank))]
    elif vranks:
        comms = LocalGroupComm.create(vranks)
        sols = [DecomposedSolver(eq, CartesianGrid(px, py, r), comm=comms[r], dtype=np.float32, device=local_rank, halo=halo)
                for r in range(vranks)]
    else:
        comm = None if dist is None else (PeerMappedComm() if args.decomp_mode == "peer" else NativeComm() if native else TorchComm())
        sols = [DecomposedSolver(eq, CartesianGrid(px, py, rank), comm=comm, dtype=np.float32, device=local_rank, halo=halo)]
    for sol in sols:
        sol.use_overlap = args.decomp_mode in ("native-overlap", "overlap", "graph")
        sol.use_graph = args.decomp_mode == "graph"
        sol.set_global_state(y0)
    sol = sols[0]
    engines = [s_.backend.engine for s_ in sols]
    eng = engines[0]
    if getattr(args, "tile_rows", 0) and on_gpu:
        for e in engines:
            e.set_tile_rows(args.tile_rows)

    def advance(nsub):
        if vranks:
            advance_group(sols, dt, nsub)  # one host thread per virtual rank
        else:
            sol.advance(dt, nsub)

    def barrier():
        for e in engines:
            e.sync()
        if dist is not None:
            import torch

            if on_gpu:
                torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        advance(substeps)
    barrier()
    launches0 = eng.stage_launches()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        advance(substeps)
    barrier()
    elapsed = my_elapsed = time.perf_counter() - t0
    launches = eng.stage_launches() - launches0
    per_rank = [my_elapsed]
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = [float(t_[0]) for t_ in allt]
        elapsed = max(per_rank)
    tile = sol.local_state()
    spot = None
    if not args.no_parity_spot:
        # EVERY rank runs the spot's substeps (they contain collectives); rank 0 alone compares its tile against the
        # C oracle run on the whole periodic field
        nspot = 7
        for s_ in sols:
            s_.set_global_state(y0)
        advance(nspot)
        for e in engines:
            e.sync()
        # ... and EVERY rank compares ITS tile with the C oracle run on the whole periodic field (each rank runs the
        # oracle itself: 7 substeps of the 4096^2 field are a fraction of a second per rank); the worst rank's errors
        # travel to rank 0 (max all-reduce), which reports them -- a wrong tile on any rank fails the run
        from oracle import c_oracle as CO

        _, cmu, cmob = _oracle_closures(WORKLOADS["ch_rk4_1024_f32"])
        ref = CO.rk4(0, y0, dom.dx[0], dom.dx[1], 0.002, cmu, cmob, dt, nspot, threads=usable_cores()).astype(np.float64)
        rel = mabs = 0.0
        for s_ in sols:  # all virtual ranks' tiles (one tile under torchrun: this rank's)
            got = s_.local_state().astype(np.float64)
            si, sj = s_.grid.tile_slices(n, n)
            base = y0[si, sj].astype(np.float64)
            e_rel = float(np.linalg.norm((got - base) - (ref[si, sj] - base)) / np.linalg.norm(ref[si, sj] - base))
            e_abs = float(np.max(np.abs(got - ref[si, sj])))
            rel = max(rel, e_rel if np.isfinite(e_rel) else np.inf)  # max(0.0, nan) is 0.0: a NaN tile must not pass
            mabs = max(mabs, e_abs if np.isfinite(e_abs) else np.inf)
        tiles_checked = len(sols)
        if dist is not None and world > 1:
            import torch

            worst = torch.tensor([rel, mabs], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
            rel, mabs = float(worst[0]), float(worst[1])
            tiles_checked = world
        spot = {"parity_spot_rel_err": rel, "parity_spot_max_abs_err": mabs,
                "parity_spot_tol": SPOT_TOL_F32["rk4"], "parity_spot_max_abs_tol": SPOT_ABS_TOL_F32,
                "parity_spot_ok": bool(rel < SPOT_TOL_F32["rk4"] and mabs < SPOT_ABS_TOL_F32),
                "parity_spot": f"all {tiles_checked} tile(s) of {sol.tile_shape[0]}x{sol.tile_shape[1]} after {nspot} substeps "
                               f"({sol.mode}), each rank against oracle/c_oracle.c on the whole periodic field (worst rank reported)"}
    if dist is not None and world > 1:
        dist.barrier()
    if rank == 0:
        total_bytes = WORDS["rk4"] * 4 * n * n * substeps * args.steps
        nex = sum(1 for f in sol.backend.phase_plan() if f >= 0)
        line = {
            "metric": "env-steps/sec (ch_rk4_4096_decomp: one 4096^2 field, 100 substeps/env-step) & achieved HBM GB/s",
            "value": args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ch_rk4_4096_decomp", "grid": [n, n], "tiles": [px, py], "tile": list(sol.tile_shape),
                       "integrator": "rk4", "dt": dt, "substeps_per_env_step": substeps, "halo": getattr(sol.backend, "halo", 4),
                       "exchange": f"one all-gather of packed halo strips per {'substep' if nex == 1 else 'fused stage pair'} ({nex} per substep)",
                       "stencil_launches_per_substep": launches / max(args.steps * substeps, 1),
                       "ranks": (f"{vranks} virtual ranks on ONE GPU (in-process group, csrc/comm.hip: the RCCL run's loop with device "
                                 "copies as the collective)") if vranks else f"{world} process(es), one GPU each",
                       "strip_bytes": int(sol.backend.strip_elems) * 4, "driver_mode": sol.mode, "kernel": eng.last_kernel},
            "substeps_per_s": args.steps * substeps / elapsed, "us_per_substep": 1e6 * elapsed / (args.steps * substeps),
            "per_rank_ms_per_step": [1e3 * t_ / args.steps for t_ in per_rank],
            "comm_world_size": (dist.get_world_size() if dist is not None else (vranks or 1)),
            "achieved_gbs_whole_job": total_bytes / elapsed / 1e9,
            "nonfinite_cells": float(np.size(tile) - np.isfinite(tile).sum()),
            **(spot or {}),
            "roofline": decomp_roofline(total_bytes / max(world, 1), elapsed, launches, sol.tile_shape, eng.last_kernel,
                                        concurrent=vranks or 1),
        }
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        if make_solver is None:
            dist.destroy_process_group()
    for e in engines:
        e.close()
    if spot is not None and not spot["parity_spot_ok"]:
        raise SystemExit(f"parity spot check FAILED: {spot}")
    return spot


def usable_cores():
    """host cores this process may use: the affinity mask, clipped by the cgroup CPU quota (a GPU box hands
    one GPU's job a share of the host, not all of os.cpu_count())"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def _oracle_closures(w):
    from oracle import c_oracle as CO

    if w["eq"] == "ch" and w.get("closures") == "cubic":
        return 0, CO.closure(0, 0, (0.0, -1.0, 0.0, 1.0)), CO.closure(0, 0, (1.0, 0.0, 1.0))
    if w["eq"] == "ch":
        return 0, CO.closure(0, 1, (3.0, -6.0)), CO.closure(0, 0, (0.0, 1.0, -1.0))
    return 1, CO.closure(0, 0, (0.0, -1.0, 0.0, 1.0)), CO.closure(0, 0, (1.0,))


def parity_spot(name, eng, eq, solver, y0, threads):
    """Post-timing parity spot check (the bench line proves its own result): the SAME library call the timed
    region makes -- pdeopt_advance on the whole batch, environment groups and all -- on the fresh inputs,
    first / last environment of the batch and the environments either side of the group edges
    against the CPU oracle: oracle/c_oracle.c for RK4 (one full environment step), oracle/np_oracle.py for
    the spectral integrators (8 substeps; it is a numpy port).  Returns the worst relative L2 error of the
    state increment (of the state for the GPE, whose norm is fixed) and the worst absolute state error."""
    from oracle import c_oracle as CO
    from oracle import np_oracle as O

    w = WORKLOADS[name]
    dt, batch = w["dt"], y0.shape[0]
    nsub = w["substeps"] if (w["integ"] == "rk4" and w["eq"] not in ("ch_sbm", "ch3d")) else (4 if w["eq"] == "ch3d" else 8)
    eng.set_state(y0)
    eng.advance(solver.integrator, dt, nsub, 0.0)
    # first / last environment of the batch and the two either side of every group edge (at most 8 environments)
    ngroups = max(1, eng.last_groups())
    gsz = -(-batch // ngroups)
    edges = {0, batch - 1, batch // 2 - 1, batch // 2}
    for g in range(1, ngroups):
        if len(edges) >= 8:
            break
        edges |= {g * gsz - 1, g * gsz}
    envs = sorted(edges & set(range(batch)))
    if w["eq"] == "ch3d":
        envs = [0, batch - 1]  # (a numpy right-hand side of 128^3 takes ~1 s)
    hx, hy = eq.domain.dx[:2]
    worst_rel = worst_abs = 0.0
    for b in envs:
        got = eng.get_state(b, 1)[0].astype(np.float64)
        if w["eq"] == "ch_sbm":
            frhs = lambda t, u: O.ch_sbm_rhs(u, eq.psi, hx, hy, 1.5, SBM_F, REGSOL, C1MC, SBM_THETA(t), SBM_FLUX(t), eq.left_half)
            ref = y0[b].astype(np.float64)
            for i in range(nsub):
                ref = O.rk4_step(frhs, i * dt, ref, dt)
        elif w["eq"] == "ch3d":
            f3 = lambda t, u: O.ch3d_rhs_fd(u, hx, hy, eq.domain.dx[2], 0.002, REGSOL, C1MC)
            ref = y0[b].astype(np.float64)
            for i in range(nsub):
                ref = O.rk4_step(f3, i * dt, ref, dt)
        elif w["integ"] == "rk4":
            code, cmu, cmob = _oracle_closures(w)
            ref = CO.rk4(code, y0[b], hx, hy, 0.002, cmu, cmob, dt, nsub, threads=threads).astype(np.float64)
        elif w["integ"] == "imex":
            sym = O.ch_fourier_symbol(w["n"], w["n"], hx, hy, 0.002)
            rhs = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, REGSOL, C1MC)
            ref = y0[b].astype(np.float64)
            for i in range(nsub):
                ref = O.imex_step(rhs, i * dt, ref, dt, 0.5, sym)
        else:
            X, Y = eq.domain.mesh()
            bt = lambda t, yy: O.gpe_b_terms(yy, X, Y, 1000.0, 0.0, 1.0, eq.lights(t, X, Y))
            ref = y0[b].astype(np.float64)
            for i in range(nsub):
                ref = O.strang_step(bt, i * dt, ref, dt, eq.A_term, eq.dx, 1.0)
        base = 0.0 if w["eq"] == "gpe" else y0[b].astype(np.float64)
        den = np.linalg.norm(ref - base)
        rel = float(np.linalg.norm((got - base) - (ref - base)) / (den if den > 0 else 1.0))
        ab = float(np.max(np.abs(got - ref)))
        # a NaN compares false with everything: a diverged run must fail the check, not slip through max()
        worst_rel = max(worst_rel, rel if np.isfinite(rel) else float("inf"))
        worst_abs = max(worst_abs, ab if np.isfinite(ab) else float("inf"))
    f64 = y0.dtype == np.float64
    tol = (SPOT_TOL_F64 if f64 else SPOT_TOL_F32)[w["integ"]]
    abs_tol = 1e-12 if f64 else w.get("abs_tol", SPOT_ABS_TOL_F32) * max(1.0, float(np.max(np.abs(y0))))
    return {"parity_spot_rel_err": worst_rel, "parity_spot_max_abs_err": worst_abs, "parity_spot_tol": tol,
            "parity_spot_max_abs_tol": abs_tol,
            "parity_spot_ok": bool(worst_rel < tol and worst_abs < abs_tol),
            "parity_spot": f"{nsub} substeps of the timed call on fresh inputs, environments {envs} of {batch} "
                           f"(groups: {eng.last_groups()}) vs "
                           + ("oracle/c_oracle.c" if (w["integ"] == "rk4" and w["eq"] not in ("ch_sbm", "ch3d")) else "oracle/np_oracle.py")}


def cpu_baseline(name, budget_s=15.0):
    """The oracle timed on the host cores, on a bounded sample of the same workload (one
    environment, as many RK4 substeps as fit the budget):
      * value: the C restatement (oracle/c_oracle.c, fused loops, OpenMP) on every core this job may use
        (usable_cores(): affinity mask clipped by the cgroup quota -- a 1-GPU box grants a 16-core share of a
        256-thread host) -- the stronger CPU baseline, so the GPU ratio is not inflated by a slow one;
      * the numpy roll-form port (oracle/np_oracle.py, what the reference's arithmetic costs when
        executed op for op, single thread) is quoted in `sample`.
    Neither is JAX-on-CPU: JAX is not installable here (no network); see BASELINE.md section 3."""
    from oracle import c_oracle as CO
    from oracle import np_oracle as O

    w = WORKLOADS[name]
    if w["integ"] != "rk4" or w["eq"] not in ("ch", "ac"):
        return None
    n, dtype, dt, substeps = w["n"], w["dtype"], w["dt"], w["substeps"]
    hx = hy = 0.01
    rng = np.random.default_rng(0)
    if w["eq"] == "ch":
        y = np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95).astype(dtype)
        f = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, REGSOL, C1MC)
        eq, cmu, cmob = 0, CO.closure(0, 1, (3.0, -6.0)), CO.closure(0, 0, (0.0, 1.0, -1.0))
    else:
        y = (0.01 * rng.standard_normal((n, n))).astype(dtype)
        f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
        eq, cmu, cmob = 1, CO.closure(0, 0, (0.0, -1.0, 0.0, 1.0)), CO.closure(0, 0, (1.0,))
    dtc = dtype(dt)
    O.rk4_step(f, 0.0, y, dtc)
    t0 = time.perf_counter()
    n_np, yy = 0, y
    while time.perf_counter() - t0 < budget_s * 0.4:
        yy = O.rk4_step(f, 0.0, yy, dtc)
        n_np += 1
    el_np = time.perf_counter() - t0
    threads = usable_cores()
    CO.rk4(eq, y, hx, hy, 0.002, cmu, cmob, dt, 2, threads=threads)  # warm-up
    chunk, n_c, yy = 8, 0, y
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s * 0.6:
        yy = CO.rk4(eq, yy, hx, hy, 0.002, cmu, cmob, dt, chunk, threads=threads)
        n_c += chunk
    el_c = time.perf_counter() - t0
    return {
        "value": (n_c / el_c) / substeps,
        "unit": "env-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"1 env of {name}: C/OpenMP oracle (oracle/c_oracle.c) {n_c} RK4 substeps in {el_c:.1f} s on "
                  f"{threads} threads = {n_c / el_c:.1f} substeps/s; numpy roll-form oracle "
                  f"(oracle/np_oracle.py) {n_np} substeps in {el_np:.1f} s on 1 thread = "
                  f"{n_np / el_np:.1f} substeps/s; host reports {os.cpu_count()} logical CPUs, {threads} usable by this job "
                  f"(affinity + cgroup quota)",
    }


# The reference's notebook-sized ADAPTIVE solves (Tsit5 + PIDController): a "step" is one diffeqsolve call over a fixed
# time span, the unit of work a trial step of the controller (accepted or rejected: 7 right-hand sides either way).
ADAPTIVE = {
    # notebooks/smooth_boundary.ipynb:228: CahnHilliard2DSmoothedBoundary 100^2, kappa 0.002, dx 0.01, theta = pi / 2,
    # PIDController(rtol 1e-4, atol 1e-6), 117 890 steps over t = 0 .. 0.1 with 200 saves upstream: a prefix of that solve
    # with the same save spacing (one save per 5e-4)
    "ch_sbm_100_tsit5": dict(n=100, dtype=np.float32, t1=2e-3, dt0=1e-6, nsave=5, rtol=1e-4, atol=1e-6),
    # notebooks/smooth_boundary.ipynb:397: the second solve of the notebook, contact angle theta(t) a quadratic in t (352 104 steps
    # upstream): the kernel evaluates cos theta at its own stage times
    "ch_sbm_100_tsit5_theta": dict(n=100, dtype=np.float32, t1=2e-3, dt0=1e-6, nsave=5, rtol=1e-4, atol=1e-6),
    "ch_sbm_100_tsit5_f64": dict(n=100, dtype=np.float64, t1=2e-3, dt0=1e-6, nsave=5, rtol=1e-4, atol=1e-6),
    # notebooks/run_advection_diffusion.ipynb:84: advection-diffusion 64^2 (8 950 steps upstream)
    "ad_64_tsit5": dict(n=64, dtype=np.float32, t1=0.4, dt0=1e-5, nsave=2, rtol=1e-4, atol=1e-6),
}


def adaptive_problem(P, name):
    import types

    w = ADAPTIVE[name]
    n = w["n"]
    if name.startswith("ch_sbm"):
        yy, xx = np.ogrid[:n, :n]
        r = np.sqrt((xx - n / 2) ** 2 + (yy - n / 2) ** 2)
        psi = np.maximum(1e-3, 0.5 * (1.0 + np.tanh((20.0 - r) / 3.0)))  # a disc of the notebook's size as a smooth level set
        dom = P.Domain((n, n), ((-0.5, 0.5), (-0.5, 0.5)), "dimensionless", geometry=types.SimpleNamespace(smooth=psi))
        theta = SBM_THETA if name.endswith("_theta") else (lambda t: np.pi / 2.0)
        eq = P.CahnHilliard2DSmoothedBoundary(dom, 0.002, SBM_F, REGSOL, C1MC, theta, lambda t: 0.0)
        y0 = 0.9 * np.ones((n, n))
        y0[:, : n // 2] = 0.1
        return eq, y0
    dom = P.Domain((n, n), ((0.0, 0.02 * n), (0.0, 0.02 * n)), "dimensionless")

    def vel(t, x, y):
        g = np.exp(-((x - 0.4) ** 2 + (y - 0.4) ** 2) / (2 * 0.01))
        return -0.1 * (x - 0.4) / 0.01 * g, -0.1 * (y - 0.4) / 0.01 * g

    return P.AdvectionDiffusion2D(dom, vel, 0.1, time_dependent=False), 0.5 + 0.01 * np.random.default_rng(0).standard_normal((n, n))


def run_adaptive(args, P):
    """--workload ch_sbm_100_tsit5 / ad_64_tsit5: the whole adaptive solve inside ONE launch (csrc/stencil_coop_adaptive.hpp:
    several workgroups per environment, controller in the kernel).  value = trial steps per second through
    ``P.diffeqsolve`` (upload, launch, statistics and save points back: everything the caller waits for)."""
    w = ADAPTIVE[args.workload]
    eq, y0 = adaptive_problem(P, args.workload)
    y0 = y0.astype(w["dtype"])
    ctl = P.PIDController(rtol=w["rtol"], atol=w["atol"])
    ts = np.linspace(0.0, w["t1"], w["nsave"])
    eng = P.HipEngine(int(os.environ.get("LOCAL_RANK", "0")))

    def solve(engine=eng, y=y0):
        return P.diffeqsolve(eq, P.Tsit5(), 0.0, w["t1"], w["dt0"], y, stepsize_controller=ctl, engine=engine, saveat=P.SaveAt(ts=ts))

    for _ in range(max(args.warmup, 1)):
        sol = solve()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sol = solve()
    elapsed = time.perf_counter() - t0
    trials = int(sol.stats["num_steps"])
    kernel = sol.stats["kernel"]
    us_per_trial = 1e6 * elapsed / (args.steps * trials)
    # the launch alone (HIP events around one more solve: upload + kernel + read-back on the engine's stream)
    eng.timer_start()
    solve()
    dev_ms = eng.timer_stop()
    shader_hz = eng.timer_clock_hz()
    n = w["n"]
    esize = np.dtype(w["dtype"]).itemsize
    sbm = args.workload.startswith("ch_sbm")
    # SURVEY 8(d)-style bytes of a trial step, had every stage gone through HBM: 7 right-hand sides x (read w, write k)
    # + the static fields per stage (psi, |grad psi| / psi, mask | two face velocities) + the combination reads
    words = 7 * (2 + (3 if sbm else 2)) + 9
    alg_bytes = words * esize * n * n
    roof = {
        "bound": "valu", "kernel": kernel, "unit": "VALU utilisation of the compute units the launch occupies",
        "achieved": None, "peak": 1.0, "frac": None, "traffic": None,
        "us_per_trial_step_wall": us_per_trial, "us_per_trial_step_device": 1e3 * dev_ms / trials,
        "algorithmic_bytes_per_trial_step": alg_bytes,
        "algorithmic_gbs": alg_bytes / (1e-3 * dev_ms / trials) / 1e9,
        "algorithmic_frac_of_hbm_peak": alg_bytes / (1e-3 * dev_ms / trials) / 1e9 / HBM_PEAK_GBS,
        "shader_clock_hz": shader_hz,
        "note": "one launch = the whole solve; the state, the six slopes and the static fields stay in LDS, HBM sees the state once in and the save points "
                "out (traffic ~ 0): the kernel is bound by the instruction issue + LDS latency of the <= 32 compute units one environment can use "
                "(one XCD: the per-step exchange then stays in one L2), not by a chip-wide roofline.  frac = VALU-busy share of those compute "
                "units' SIMD cycles from the committed PMC pass of this command (null while none is committed).",
    }
    try:
        pmc = json.load(open(PMC_FILE)).get(args.workload)
    except Exception:
        pmc = None
    m = __import__("re").search(r"(\d+)x(\d+) workgroups", kernel)
    nwg = int(m.group(1)) * int(m.group(2)) if m else None
    if pmc and nwg:
        c = pmc["counters_per_launch"]
        hz = shader_hz if shader_hz and shader_hz > 1e8 else SHADER_HZ
        cycles = 1e-3 * dev_ms * hz
        if c.get("SQ_ACTIVE_INST_VALU"):
            roof["workgroups"] = nwg
            roof["valu_insts_per_trial_step"] = c.get("SQ_INSTS_VALU", 0) / trials
            roof["achieved"] = roof["frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (4 * nwg) / cycles / VALU_BUSY_SATURATED
            roof["traffic"] = pmc.get("hbm_bytes_per_launch")
            roof["pmc_source"] = os.path.relpath(PMC_FILE, ROOT) + ": " + pmc.get("source", "")
    # parity + CPU baseline in one: the same solve driven step by step on the numpy oracle under the package's own host loop
    spot, cpu = {}, None
    if not args.no_parity_spot or not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from fake_engine import OracleEngine

        tc = time.perf_counter()
        ref = solve(engine=OracleEngine(), y=y0.astype(np.float64))
        el_c = time.perf_counter() - tc
        n_ref = int(ref.stats["num_steps"])
        got, want = sol.ys.astype(np.float64), ref.ys
        mabs = float(np.max(np.abs(got - want)))
        same = (int(sol.stats["num_accepted_steps"]), int(sol.stats["num_rejected_steps"])) == (int(ref.stats["num_accepted_steps"]), int(ref.stats["num_rejected_steps"]))
        tol = (1e-9 if esize == 8 else 2e-5) if same else 2e-3
        spot = {"parity_spot_max_abs_err": mabs, "parity_spot_tol": tol, "parity_spot_ok": bool(np.isfinite(mabs) and mabs < tol),
                "parity_spot_same_accept_reject_sequence": bool(same),
                "parity_spot": f"the same solve driven step by step on oracle/np_oracle.py in fp64 ({n_ref} trial steps; this run {trials}): "
                               f"largest difference at the save points"}
        cpu = {"value": n_ref / el_c, "unit": "trial-steps/s", "cores": 1, "kind": "port",
               "sample": f"the whole solve ({n_ref} trial steps, 7 right-hand sides each) on the numpy oracle, 1 thread, {el_c:.1f} s"}
    line = {
        "metric": f"adaptive trial steps/sec ({args.workload}: Tsit5 + PID rtol {w['rtol']:g} atol {w['atol']:g}, t = 0 .. {w['t1']:g})",
        "value": args.steps * trials / elapsed, "unit": "trial-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if esize == 4 else "f64", "data": "synthetic",
        "config": {"workload": args.workload, "grid": [n, n], "envs_per_gpu": 1, "integrator": "tsit5 + PID controller (in-kernel)",
                   "trial_steps_per_solve": trials, "accepted": int(sol.stats["num_accepted_steps"]), "rejected": int(sol.stats["num_rejected_steps"]),
                   "kernel": kernel, "step": "one P.diffeqsolve call over the span: upload, ONE launch, statistics + save points back"},
        "us_per_trial_step": us_per_trial,
        "roofline": roof,
        "cpu_baseline": cpu,
        **spot,
    }
    eng.close()
    print(json.dumps(line))
    sys.stdout.flush()
    if spot and not spot["parity_spot_ok"]:
        raise SystemExit(f"adaptive workload FAILED its parity spot: {spot}")
    return line


def run_single_process(args, P, engines=None):
    """--single-process: the OTHER way to use the GPUs of a node -- ONE process,
    ``VectorPDEEnv(devices=[0..N-1])``: one engine + one dedicated host thread per device, environments sharded by
    contiguous blocks, no collective.  A step = ``VectorPDEEnv.step`` of all N x batch environments (per-environment
    kappa control, variance reward reduced on each device; the states never leave HBM).  Same JSON line as the
    process-per-GPU mode, ``per_rank_ms_per_step`` = each DEVICE's own wall time per step.  ``engines``: test hook (the CPU
    suite passes oracle-backed engine doubles in place of ``devices=``)."""
    w = WORKLOADS[args.workload]
    if w["eq"] not in ("ch", "ac") or w["integ"] not in ("rk4", "imex"):
        raise SystemExit("--single-process runs the Cahn-Hilliard / Allen-Cahn workloads")
    n, batch, N = w["n"], args.batch_per_gpu or w["batch"], args.gpus
    L_ = 0.01 * n
    dom = P.Domain((n, n), ((-L_ / 2, L_ / 2), (-L_ / 2, L_ / 2)), "dimensionless")

    def reset(domain, seed=0):
        rng = np.random.default_rng(seed)
        if w["eq"] == "ch":
            return np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95).astype(w["dtype"])
        return (0.01 * rng.standard_normal((n, n))).astype(w["dtype"])

    if w["eq"] == "ch":
        cubic = w.get("closures") == "cubic"
        eq_t, static = P.CahnHilliard2DPeriodic, {"mu": (lambda c: c**3 - c) if cubic else REGSOL, "D": (lambda c: 1 + c**2) if cubic else C1MC}
    else:
        eq_t, static = P.AllenCahn2DPeriodic, {"mu": lambda c: c**3 - c, "R": lambda c: np.ones_like(c)}
    imex = w["integ"] == "imex"
    total = batch * N
    actions = [1] * total  # kappa stays 0.002: the timed arithmetic is the workload's own (the control path still runs per step)
    env = P.VectorPDEEnv(
        total, eq_t, dom, P.SemiImplicitFourierSpectral if imex else P.RK4, end_time=1e9, step_dt=w["dt"] * w["substeps"],
        numeric_dt=w["dt"], state_to_observation_func=lambda s_: s_, reward_function=lambda s_: 0.0, reset_func=reset,
        reset_control_value=0.002, update_control_value=lambda off, old: old + off, update_control_parameter=lambda old, new: new,
        action_space_config={"type": "discrete", "num_actions": 3, "action_mapping": {0: -1e-5, 1: 0.0, 2: 1e-5}},
        static_equation_parameters=static, control_equation_parameter_name="kappa", solver_parameters={"A": 0.5} if imex else {},
        device_reward="var", fetch_observations=False, **({"devices": list(range(N))} if engines is None else {"engines": engines}))
    env.reset(seed=0)
    for _ in range(args.warmup):
        env.step(actions)
    per_dev = np.zeros(N)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, rew, *_ = env.step(actions)  # returns after every device's reward reduction: all streams are drained
        per_dev += np.asarray(env.last_step_seconds)
    elapsed = time.perf_counter() - t0
    kernel = env._shards[0].engine.last_kernel
    # parity: the first and last environment of the job against the C oracle over the steps taken (RK4 workloads)
    spot = None
    if not args.no_parity_spot and w["integ"] == "rk4" and w["dtype"] is np.float32:
        from oracle import c_oracle as CO

        eq_id, cmu, cmob = _oracle_closures(w)
        st = env.states
        nsteps = (args.warmup + args.steps) * w["substeps"]
        rel = mabs = 0.0
        for b in (0, total - 1):
            y0 = reset(dom, seed=b)
            ref = CO.rk4(eq_id, y0, dom.dx[0], dom.dx[1], 0.002, cmu, cmob, w["dt"], nsteps, threads=usable_cores()).astype(np.float64)
            e_rel = float(np.linalg.norm((st[b] - y0.astype(np.float64)) - (ref - y0)) / np.linalg.norm(ref - y0))
            e_abs = float(np.max(np.abs(st[b] - ref)))
            rel = max(rel, e_rel if np.isfinite(e_rel) else np.inf)
            mabs = max(mabs, e_abs if np.isfinite(e_abs) else np.inf)
        tol = SPOT_TOL_F32["rk4"] * max(1.0, (args.warmup + args.steps) / 2)  # rounding accumulates with the steps taken
        spot = {"parity_spot_rel_err": rel, "parity_spot_max_abs_err": mabs, "parity_spot_tol": tol,
                "parity_spot_ok": bool(rel < tol and np.isfinite(mabs)),
                "parity_spot": f"environments 0 and {total - 1} after {nsteps} substeps vs oracle/c_oracle.c"}
    line = {
        "metric": METRIC if args.workload == "ch_rk4_1024_f32" else f"env-steps/sec ({args.workload}, {w['substeps']} substeps/env-step) & achieved HBM GB/s",
        "value": total * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if np.dtype(w["dtype"]).itemsize == 4 else "f64", "data": "synthetic",
        "config": {"workload": args.workload, "grid": [n, n], "envs_per_gpu": batch, "envs_total": total, "integrator": w["integ"],
                   "dt": w["dt"], "substeps_per_env_step": w["substeps"], "kernel": kernel,
                   "mode": "single process: VectorPDEEnv(devices=[0..N-1]), one engine + one host thread per device, no collective",
                   "step": "VectorPDEEnv.step: per-environment kappa control, one pdeopt_advance per device, variance reward reduced on the device"},
        "per_rank_ms_per_step": [1e3 * float(t_) / args.steps for t_ in per_dev],
        "shard_bounds": env.shard_bounds,
        "nonfinite_rewards": int(np.size(rew) - np.isfinite(rew).sum()),
        **(spot or {}),
    }
    env.close()
    print(json.dumps(line))
    sys.stdout.flush()
    if line["nonfinite_rewards"] or (spot is not None and not spot["parity_spot_ok"]):
        raise SystemExit(f"single-process run FAILED its checks: {spot}, non-finite rewards {line['nonfinite_rewards']}")
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ch_rk4_1024_f32", choices=sorted(WORKLOADS) + sorted(ADAPTIVE) + ["ch_rk4_4096_decomp"])
    ap.add_argument("--batch-per-gpu", type=int, default=0)
    ap.add_argument("--kernel-path", type=int, default=0, help="0 auto, 1 generic, 2 tiled")
    ap.add_argument("--tile-rows", type=int, default=0, help="0 auto, 16 or 32")
    ap.add_argument("--group-envs", type=int, default=0, help="environments per cache-resident group (0 auto, -1 whole batch)")
    ap.add_argument("--fuse", type=int, default=0, help="RK4 stage-pair fusion: 0 auto, -1 off")
    ap.add_argument("--group-streams", type=int, default=0, help="environment groups: 0 auto (two side by side on two streams), 1 one at a time")
    ap.add_argument("--ablate", type=int, default=0, help="TIMING ONLY (wrong results): kernel phase ablation bits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-spot", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the VectorPDEEnv.step throughput leg")
    ap.add_argument("--decomp-mode", default="auto", choices=["auto", "native", "native-overlap", "peer", "plain", "overlap", "graph"],
                    help="ch_rk4_4096_decomp: driver path (auto = native: the library's own RCCL communicator, substep "
                         "loop in C; native-overlap: + collective on a second stream under the interior tiles; "
                         "peer = no collective: neighbours' strips read in place through hipIpc-mapped buffers; "
                         "plain / overlap / graph = torch all-gather drivers)")
    ap.add_argument("--decomp-grid", type=int, default=4096, help="ch_rk4_4096_decomp: cells per side of the one field")
    ap.add_argument("--decomp-substeps", type=int, default=100, help="ch_rk4_4096_decomp: substeps per step")
    ap.add_argument("--decomp-halo", type=int, default=0, choices=[0, 4, 8],
                    help="ch_rk4_4096_decomp: halo layout (0 = auto: 8 = one exchange per substep, 4 for the overlap / graph drivers)")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="ch_rk4_4096_decomp: run this many ranks of ONE process on ONE GPU (in-process group)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives all --gpus N devices through VectorPDEEnv(devices=[0..N-1]) (one engine + one host "
                         "thread per device) instead of one process per GPU under torch.distributed")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):  # launched by torchrun
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        # RCCL printf()s its version banner to stdout when the communicator is created (NCCL_DEBUG=VERSION
        # on the GPU boxes): create it with fd 1 pointed at stderr so stdout carries the one JSON line only
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import pde_opt_amd as P
    from pde_opt_amd import _lib as L

    if args.single_process:
        if world > 1:
            raise SystemExit("--single-process is ONE process: start it with `python bench.py --gpus N --single-process`, not under torchrun")
        return run_single_process(args, P)
    if args.workload == "ch_rk4_4096_decomp":
        return run_decomp(args, P, world, rank, local_rank, dist)
    if args.workload in ADAPTIVE:
        if world > 1:
            raise SystemExit("the adaptive workloads are single-environment latency measurements: one GPU")
        return run_adaptive(args, P)
    w = WORKLOADS[args.workload]
    batch = args.batch_per_gpu or w["batch"]
    eq, y0, solver = make_problem(P, args.workload, batch, rank)
    dt, substeps = w["dt"], w["substeps"]
    eng = P.HipEngine(local_rank)  # fails loudly without the HIP library / a GPU
    eng.set_kernel_path(args.kernel_path)
    eng.set_tile_rows(args.tile_rows)
    eng.set_group_envs(args.group_envs)
    eng.set_fuse_stages(args.fuse)
    eng.set_group_streams(args.group_streams)
    if args.ablate:
        eng._check(eng._lib.pdeopt_set_option(eng._h, L.OPT_DEBUG_ABLATE, args.ablate))
    eng.configure(dtype=y0.dtype, batch=batch, **eq._engine_problem())
    eq._engine_upload(eng, 0.0, dt * substeps)
    solver.configure_engine(eng, eq)
    eng.set_state(y0)  # inputs resident in HBM before the timed region

    def env_step():
        eng.advance(solver.integrator, dt, substeps, 0.0)

    def barrier():
        eng.sync()
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        env_step()
    barrier()
    launches0 = eng.stage_launches()
    eng.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        env_step()
    dev_ms = eng.timer_stop()  # HIP events on the engine's stream (synchronises)
    barrier()
    elapsed = time.perf_counter() - t0
    shader_hz = eng.timer_clock_hz()  # the clock the chip held between the two events (fetched outside the timed region)
    kernel_name = eng.last_kernel
    launches = eng.stage_launches() - launches0  # fused stencil(+update) launches in the timed region

    per_rank = [elapsed]
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device="cuda")
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)  # every rank's clock: the line reports them all, value uses the slowest
        per_rank = [float(t_[0]) for t_ in allt]
        elapsed, dev_ms = max(per_rank), max(float(t_[1]) for t_ in allt)

    # sanity: the timed state is finite (a diverged run would be measuring NaN arithmetic)
    bad = float(eng.reduce(L.RED_NONFINITE).sum())
    spot = None
    if rank == 0 and not args.no_parity_spot and not args.ablate:
        spot = parity_spot(args.workload, eng, eq, solver, y0, usable_cores())

    if rank == 0:
        nx, ny = eq.domain.points[:2]
        esize = y0.dtype.itemsize
        words = WORDS[w.get("words", w["integ"])]
        total_bytes = words * esize * int(np.prod(eq.domain.points)) * batch * substeps * args.steps
        launches = max(launches, 1)
        bytes_per_launch = total_bytes / launches
        # two environment groups side by side (PDEOPT_OPT_GROUP_STREAMS): a launch lasts twice its share of the timed region
        concurrent = eng.last_group_streams()
        avg_launch_s = (dev_ms * 1e-3) * concurrent / launches
        line = {
            "metric": METRIC if args.workload == "ch_rk4_1024_f32" else
                      f"env-steps/sec ({args.workload}, {substeps} substeps/env-step) & achieved HBM GB/s",
            "value": args.gpus * batch * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"gpe": "c64"}.get(w["eq"], "f32" if esize == 4 else "f64"),
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "grid": [int(v) for v in eq.domain.points],
                "envs_per_gpu": batch,
                "envs_total": batch * args.gpus,
                "integrator": w["integ"],
                "dt": dt,
                "substeps_per_env_step": substeps,
                "sharding": "independent environments per GPU, no collective in the step",
                "kernel": kernel_name,
            },
            "substeps_per_s": args.gpus * batch * args.steps * substeps / elapsed,
            "per_rank_ms_per_step": [1e3 * t_ / args.steps for t_ in per_rank],  # value uses the slowest rank
            "comm_world_size": dist.get_world_size() if dist is not None else 1,  # as RCCL / torch.distributed sees it
            "achieved_gbs_whole_job": args.gpus * total_bytes / elapsed / 1e9,
            "nonfinite_cells": bad,
            **(spot or {}),
            "roofline": roofline_block(args.workload, kernel_name, bytes_per_launch, words, avg_launch_s, launches, concurrent, shader_hz),
        }
        if args.gpus == 1 and not args.no_api and not args.ablate:
            line.update(api_throughput(P, args.workload, rank, args.steps, args.warmup) or {})
        if args.gpus == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload, args.cpu_seconds)
            if cb is not None:
                line["cpu_baseline"] = cb
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if spot is not None and not spot["parity_spot_ok"]:
        raise SystemExit(f"parity spot check FAILED: {spot}")
    if bad > 0:
        raise SystemExit(f"the timed state holds {int(bad)} non-finite cells: the run measured NaN arithmetic")


if __name__ == "__main__":
    main()
