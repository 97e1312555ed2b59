"""BASELINE config 5: ONE Cahn-Hilliard field of grid^2 cells, RK4, decomposed over px x py GPUs with an
RCCL all-gather of packed halo strips per fused stage pair (pde_opt_amd/decomp.py).

  one GPU (monolithic tile, the exchange is the library's loopback):
      python tools/decomp_bench.py --grid 4096
  2 x 2 GPUs of one node:
      python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 \
          tools/decomp_bench.py --grid 4096 --px 2 --py 2

Prints one JSON line on rank 0: RK4 substeps/s of the whole field and the algorithmic GB/s of SURVEY 8(d)
(64 B per cell and substep at fp32)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--px", type=int, default=1)
    ap.add_argument("--py", type=int, default=1)
    ap.add_argument("--substeps", type=int, default=100)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.px * args.py:
        raise SystemExit(f"process grid {args.px}x{args.py} needs {args.px * args.py} ranks, WORLD_SIZE={world}")
    dist = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import pde_opt_amd as P
    from pde_opt_amd.decomp import CartesianGrid, DecomposedSolver, TorchComm

    n = args.grid
    dom = P.Domain((n, n), ((-0.005 * n, 0.005 * n),) * 2, "dimensionless")
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c), lambda c: c * (1 - c))
    grid = CartesianGrid(args.px, args.py, rank)
    comm = TorchComm() if dist is not None else None
    sol = DecomposedSolver(eq, grid, comm=comm, dtype=np.float32, device=local_rank)
    # every rank draws the same global field and keeps its tile (seeded: SURVEY 8(d) config 5)
    rng = np.random.default_rng(0)
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95).astype(np.float32)
    sol.set_global_state(y0)
    del y0

    def sync():
        sol.backend.engine.sync()
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        sol.advance(2e-7, args.substeps)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sol.advance(2e-7, args.substeps)
    sync()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch

        tt = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt[0])
    tile = sol.local_state()
    ok = bool(np.isfinite(tile).all())
    if rank == 0:
        sub = args.steps * args.substeps / el
        print(json.dumps({
            "workload": f"ch_rk4_{n}_f32 single field, {args.px}x{args.py} tiles", "n_gpus": world,
            "substeps_per_s": sub, "env_steps_per_s": sub / args.substeps, "us_per_substep": 1e6 / sub,
            "algorithmic_gbs": 64.0 * n * n * sub / 1e9, "exchanges_per_substep": sol.exchanges / ((args.steps + args.warmup) * args.substeps),
            "strip_bytes": int(sol.backend.strip_elems) * 4, "finite": ok,
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
