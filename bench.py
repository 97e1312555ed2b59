"""bench.py -- env-steps/sec and achieved HBM GB/s of the fused Cahn-Hilliard RK4 step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): env-steps/sec & achieved HBM GB/s, Cahn-Hilliard 1024^2 RK4.
Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  1024^2 fp32 field, kappa=0.002, mu = log(c/(1-c)) + 3(1-2c), D = c(1-c), IC
  clip(0.5 + 0.01 N(0,1), 0.05, 0.95) seeded per environment, explicit RK4 dt=2e-7,
  100 substeps per environment step, 32 environments per GPU (256 over 8 GPUs).
A "step" = one environment step of every environment on the rank = 100 RK4 substeps
(400 fused stencil+update launches).  Weak scaling: per-GPU work is fixed; environments are
independent, there is no data-path collective.

Inputs are resident in HBM when the timed region starts (states uploaded before warm-up).
Timing: barrier + device sync on both sides, max over ranks; the roofline figure is measured live
with HIP events recorded on the engine's own stream (pdeopt_timer_start/stop).
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
RK4_BYTES_PER_CELL_SUBSTEP = {4: 64, 8: 128}  # 16 words: SURVEY 8(d)

WORKLOADS = {
    # name: (equation, nx, ny, dtype, dt, substeps, default batch per GPU)
    "ch_rk4_1024_f32": ("ch", 1024, 1024, np.float32, 2e-7, 100, 32),
    "ac_rk4_512_f32": ("ac", 512, 512, np.float32, 5e-5, 100, 64),
    "ch_rk4_1024_f64": ("ch", 1024, 1024, np.float64, 2e-7, 100, 16),
}


def make_problem(P, name, batch, rank):
    kind, nx, ny, dtype, dt, substeps, _ = WORKLOADS[name]
    lx, ly = 0.01 * nx, 0.01 * ny
    dom = P.Domain((nx, ny), ((-lx / 2, lx / 2), (-ly / 2, ly / 2)), "dimensionless")
    if kind == "ch":
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c), lambda c: c * (1 - c))
    else:
        eq = P.AllenCahn2DPeriodic(dom, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
    y0 = np.empty((batch, nx, ny), dtype=dtype)
    for b in range(batch):
        rng = np.random.default_rng(rank * batch + b)  # seeds 0..255 over 8 GPUs x 32 envs
        if kind == "ch":
            y0[b] = np.clip(0.5 + 0.01 * rng.standard_normal((nx, ny)), 0.05, 0.95)
        else:
            y0[b] = 0.01 * rng.standard_normal((nx, ny))
    return eq, y0, dt, substeps


def cpu_baseline(name, budget_s=15.0):
    """The oracle timed on the host cores, on a bounded sample of the same workload (one
    environment, as many RK4 substeps as fit the budget):
      * value: the C restatement (oracle/c_oracle.c, fused loops, OpenMP) on min(16, cores) threads
        -- the stronger CPU baseline, so the GPU ratio is not inflated by a slow one;
      * the numpy roll-form port (oracle/np_oracle.py, what the reference's arithmetic costs when
        executed op for op, single thread) is quoted in `sample`.
    Neither is JAX-on-CPU: JAX is not installable here (no network); see BASELINE.md section 3."""
    from oracle import c_oracle as CO
    from oracle import np_oracle as O

    kind, nx, ny, dtype, dt, substeps, _ = WORKLOADS[name]
    hx = hy = 0.01
    rng = np.random.default_rng(0)
    if kind == "ch":
        mu = lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c)
        mob = lambda c: c * (1 - c)
        y = np.clip(0.5 + 0.01 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(dtype)
        f = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, mu, mob)
        eq, cmu, cmob = 0, CO.closure(0, 1, (3.0, -6.0)), CO.closure(0, 0, (0.0, 1.0, -1.0))
    else:
        y = (0.01 * rng.standard_normal((nx, ny))).astype(dtype)
        f = lambda t, u: O.ac_rhs_fd(u, hx, hy, 0.002, lambda c: c**3 - c, lambda c: np.ones_like(c))
        eq, cmu, cmob = 1, CO.closure(0, 0, (0.0, -1.0, 0.0, 1.0)), CO.closure(0, 0, (1.0,))
    dtc = dtype(dt)
    # numpy roll-form port, single thread
    O.rk4_step(f, 0.0, y, dtc)
    t0 = time.perf_counter()
    n_np = 0
    yy = y
    while time.perf_counter() - t0 < budget_s * 0.4:
        yy = O.rk4_step(f, 0.0, yy, dtc)
        n_np += 1
    el_np = time.perf_counter() - t0
    # C / OpenMP port
    threads = max(1, min(16, os.cpu_count() or 1))
    CO.rk4(eq, y, hx, hy, 0.002, cmu, cmob, dt, 2, threads=threads)  # warm-up
    chunk, n_c = 8, 0
    t0 = time.perf_counter()
    yy = y
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
                  f"{n_np / el_np:.1f} substeps/s; host reports {os.cpu_count()} cores",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ch_rk4_1024_f32", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-per-gpu", type=int, default=0)
    ap.add_argument("--kernel-path", type=int, default=0, help="0 auto, 1 generic, 2 tiled")
    ap.add_argument("--tile-rows", type=int, default=0, help="0 auto (16), 16 or 32")
    ap.add_argument("--group-envs", type=int, default=0, help="environments per cache-resident group (0 auto, -1 whole batch)")
    ap.add_argument("--fuse", type=int, default=0, help="RK4 stage-pair fusion: 0 auto, -1 off")
    ap.add_argument("--ablate", type=int, default=0, help="TIMING ONLY (wrong results): kernel phase ablation bits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
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
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import pde_opt_amd as P
    from pde_opt_amd import _lib as L

    batch = args.batch_per_gpu or WORKLOADS[args.workload][6]
    eq, y0, dt, substeps = make_problem(P, args.workload, batch, rank)
    eng = P.HipEngine(local_rank)  # fails loudly without the HIP library / a GPU
    eng.set_kernel_path(args.kernel_path)
    eng.set_tile_rows(args.tile_rows)
    eng.set_group_envs(args.group_envs)
    eng.set_fuse_stages(args.fuse)
    if args.ablate:
        eng._check(eng._lib.pdeopt_set_option(eng._h, L.OPT_DEBUG_ABLATE, args.ablate))
    eng.configure(dtype=y0.dtype, batch=batch, **eq._engine_problem())
    eng.set_state(y0)  # inputs resident in HBM before the timed region

    def env_step():
        eng.advance(L.INT_RK4, dt, substeps, 0.0)

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
    kernel_name = eng.last_kernel

    if dist is not None:
        import torch

        tt = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(tt[0]), float(tt[1])

    # sanity: the timed state is finite (a diverged run would be measuring NaN arithmetic)
    bad = float(eng.reduce(L.RED_NONFINITE).sum())

    if rank == 0:
        nx, ny = eq.domain.points
        esize = y0.dtype.itemsize
        launches = eng.stage_launches() - launches0  # fused stage launches in the timed region
        total_bytes = RK4_BYTES_PER_CELL_SUBSTEP[esize] * nx * ny * batch * substeps * args.steps
        bytes_per_launch = total_bytes / launches
        avg_launch_s = (dev_ms * 1e-3) / launches
        achieved = bytes_per_launch / avg_launch_s / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "env-steps/sec (Cahn-Hilliard 1024^2 RK4, 100 substeps/env-step) & achieved HBM GB/s",
            "value": args.gpus * batch * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if esize == 4 else "f64",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "grid": [nx, ny],
                "envs_per_gpu": batch,
                "envs_total": batch * args.gpus,
                "integrator": "RK4 explicit",
                "dt": dt,
                "substeps_per_env_step": substeps,
                "sharding": "independent environments per GPU, no collective in the step",
                "kernel": kernel_name,
            },
            "substeps_per_s": args.gpus * batch * args.steps * substeps / elapsed,
            "achieved_gbs_whole_job": args.gpus * bytes_per_launch * launches / elapsed / 1e9,
            "nonfinite_cells": bad,
            "roofline": {
                "bound": "hbm",
                "kernel": "stage_tiled_kernel (average over the 4 RK4 stage launches of a substep)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_copy_6290": achieved / 6290.0,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_us": avg_launch_s * 1e6,
                "launches_timed": launches,
            },
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
