"""world_size-2 and -4 gloo tests (CPU): the N>1 code paths -- environment sharding + result gather
for batched episodes, and the halo-exchange protocol of the domain-decomposed solver -- with the
compute step supplied by the oracle (the HIP backend needs a GPU)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

from decomp_util import OracleTileBackend
from oracle import np_oracle as O
from pde_opt_amd.decomp import CartesianGrid, DecomposedSolver, TorchComm
from pde_opt_amd.sharding import gather_per_env, shard_envs
from util import MOB, MU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


class _Eq:  # the few attributes DecomposedSolver reads
    class domain:
        points = None


def _decomp_worker(rank, world, port, px, py, nx, ny, y0, want, q):
    try:
        _init(rank, world, port)
        grid = CartesianGrid(px, py, rank)
        eq = _Eq()
        eq.domain = type("D", (), {"points": (nx, ny)})
        tile = grid.tile_shape(nx, ny)
        backend = OracleTileBackend(tile, 0.01, 0.01, 0.002, MU["regsol"], MOB["c1mc"])
        s = DecomposedSolver(eq, grid, comm=TorchComm(), dtype=np.float64, backend=backend)
        s.set_global_state(y0)
        s.advance(2e-7, 3)
        si, sj = grid.tile_slices(nx, ny)
        err = float(np.max(np.abs(s.local_state() - want[si, sj])))
        q.put((rank, err, s.exchanges))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e), -1))
        raise


@pytest.mark.parametrize("px,py", [(2, 1), (1, 2), (2, 2)])
def test_halo_exchange_protocol_gloo(px, py):
    world = px * py
    rng = np.random.default_rng(3)
    nx, ny = 16 * px, 24 * py
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny)), 0.05, 0.95)
    f = lambda t, u: O.ch_rhs_fd(u, 0.01, 0.01, 0.002, MU["regsol"], MOB["c1mc"])
    want = y0
    for _ in range(3):
        want = O.rk4_step(f, 0.0, want, 2e-7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_decomp_worker, args=(r, world, port, px, py, nx, ny, y0, want, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, err, nex in sorted(res):
        assert nex == 12, (rank, err)
        # same per-cell arithmetic as the monolithic oracle: the decomposition changes nothing
        assert err == 0.0, (rank, err)


def _shard_worker(rank, world, port, total, q):
    try:
        _init(rank, world, port)
        lo, hi = shard_envs(total, world, rank)
        local = np.arange(lo, hi, dtype=np.float64) ** 2  # a per-environment scalar (e.g. a reward)
        allv = gather_per_env(local, total)
        q.put((rank, lo, hi, allv.tolist()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, -1, -1, repr(e)))
        raise


@pytest.mark.parametrize("total", [8, 7])
def test_env_sharding_and_gather_gloo(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    covered = []
    for rank, lo, hi, allv in res:
        covered += list(range(lo, hi))
        assert allv == [float(i) ** 2 for i in range(total)]
    assert covered == list(range(total))


def test_shard_envs_properties():
    for total in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            parts = [shard_envs(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


class _NullEngine:
    """what bench.run_decomp asks of a tile backend's engine (the oracle tile has none)"""
    last_kernel = "oracle tile (test double)"

    def sync(self):
        pass

    def stage_launches(self):
        return 0

    def close(self):
        pass


def _bench_decomp_worker(rank, world, port, q):
    try:
        import argparse
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        import pde_opt_amd as P

        _init(rank, world, port)

        def make_solver(eq, grid):
            nx, ny = eq.domain.points
            be = OracleTileBackend(grid.tile_shape(nx, ny), *eq.domain.dx, 0.002, bench.REGSOL, bench.C1MC, dtype=np.float32)
            be.engine, be.halo = _NullEngine(), 4
            return DecomposedSolver(eq, grid, comm=TorchComm(), dtype=np.float32, backend=be)

        args = argparse.Namespace(decomp_grid=32, decomp_substeps=3, virtual_ranks=0, decomp_mode="plain", decomp_halo=0,
                                  warmup=1, steps=2, no_parity_spot=False)
        spot = bench.run_decomp(args, P, world, rank, 0, dist, make_solver=make_solver)
        q.put((rank, None if spot is None else spot["parity_spot_ok"], None if spot is None else spot["parity_spot_rel_err"]))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:  # pragma: no cover
        q.put((rank, repr(e), -1))
        raise


def test_bench_decomp_control_flow_gloo():
    """bench.py --workload ch_rk4_4096_decomp under world_size 2: the post-timing parity spot runs
    collectives, so EVERY rank must execute its substeps -- with them under `if rank == 0` the other rank sits in the
    barrier and the job hangs (this test then times out) -- and every rank checks its own tile against the C oracle
    (round 3 compared rank 0's only).  Oracle-backed tiles, gloo all-gather, 32^2 field."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_decomp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=180) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    # every rank compared ITS tile with the C oracle and all hold the worst rank's error (max all-reduce)
    assert res[0][1] is True and res[0][2] < 1e-5, res
    assert res[1][1] is True and res[1][2] == res[0][2], res
