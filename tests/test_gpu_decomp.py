"""Domain decomposition on the GPU: a decomposed periodic field must reproduce the monolithic
periodic solve BITWISE (explicit FD: same arithmetic per cell)."""
import numpy as np
import pytest

import pde_opt_amd as P
from decomp_util import InProcessComm, gather_all
from pde_opt_amd import _lib as L
from pde_opt_amd.decomp import CartesianGrid, DecomposedSolver, HipTileBackend, LocalGroupComm, advance_group
from util import MOB, MU, std_domain

pytestmark = pytest.mark.gpu


def _monolithic(eq, y0, dt, n, fuse):
    eng = P.HipEngine()
    eng.set_fuse_stages(fuse)
    eng.set_small_persist(-1)  # the tiled kernels (a single mid-sized environment would otherwise take several CUs: other association)
    out = P.diffeqsolve(eq, P.RK4(), 0.0, n * dt, dt, y0, engine=eng).ys[-1]
    kern = eng.last_kernel
    eng.close()
    return out, kern


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("fuse,halo", [(0, 8), (1, 8), (0, 4), (-1, 4)])
def test_single_rank_loopback_equals_periodic(dtype, fuse, halo):
    rng = np.random.default_rng(0)
    dom = std_domain(P, 64, 128)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((64, 128)), 0.05, 0.95).astype(dtype)
    want, kern = _monolithic(eq, y0, 2e-7, 6, fuse)
    assert ("pair" in kern or "rk4_quad" in kern) == (fuse >= 0)  # fp32, fuse 0: the monolithic run is the whole-substep kernel
    backend = HipTileBackend(eq, (64, 128), dtype, halo=halo)
    backend.engine.set_fuse_stages(fuse)
    s = DecomposedSolver(eq, CartesianGrid(1, 1, 0), dtype=dtype, backend=backend)
    plan = backend.phase_plan()
    # halo 8: the whole-substep kernel where it applies (fp32, not asked for the pairs), else the two stage pairs
    quad = halo == 8 and fuse == 0 and dtype is np.float32
    assert plan == ([0] if quad else [0, -1] if halo == 8 else ([0, 2] if fuse == 0 else [0, 1, 2, 1]))
    s.set_global_state(y0)
    s.advance(2e-7, 6)
    assert ("rk4_quad<f32,CH,halo8" in backend.engine.last_kernel) == quad, backend.engine.last_kernel
    np.testing.assert_array_equal(s.local_state(), want)
    assert s.exchanges == 6 * sum(1 for f in plan if f >= 0)


def test_halo_layout_selection():
    """halo=None picks 8 where the fused Cahn-Hilliard stage pairs run, 4 otherwise; an impossible request is loud"""
    dom = std_domain(P, 64, 128)
    ch = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    ac = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"])
    assert HipTileBackend(ch, (64, 128), np.float32).halo == 8
    assert HipTileBackend(ac, (64, 128), np.float32).halo == 4
    with pytest.raises(ValueError, match="halo=8 needs"):
        HipTileBackend(ac, (64, 128), np.float32, halo=8)
    s = DecomposedSolver(ch, CartesianGrid(1, 1, 0), dtype=np.float32)
    s.use_overlap = True
    with pytest.raises(ValueError, match="halo-4 layout"):
        s.advance(2e-7, 2)


@pytest.mark.parametrize("grid", [(2, 2), (2, 1), (1, 4)])
@pytest.mark.parametrize("fuse,halo", [(0, 8), (1, 8), (0, 4), (-1, 4)])
def test_multi_tile_on_one_gpu_equals_monolithic(grid, fuse, halo):
    """px x py ranks played by px*py engines on one GPU; strips cross between engines through the
    same pack / unpack kernels and neighbour tables the RCCL path uses."""
    px, py = grid
    rng = np.random.default_rng(1)
    nx, ny = 64 * px, 128 * py
    dom = std_domain(P, nx, ny)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(np.float32)
    want, _ = _monolithic(eq, y0, 2e-7, 4, fuse)
    comm = InProcessComm(px * py)
    solvers = []
    for r in range(px * py):
        be = HipTileBackend(eq, (64, 128), np.float32, halo=halo)
        be.engine.set_fuse_stages(fuse)
        s = DecomposedSolver(eq, CartesianGrid(px, py, r), comm=comm.view(r), dtype=np.float32, backend=be)
        s.set_global_state(y0)
        solvers.append(s)
    plan = solvers[0].backend.phase_plan()
    for _ in range(4):
        for phase, field in enumerate(plan):
            if field >= 0:
                for s in solvers:
                    s.backend.pack(field, s.send)
                gather_all(comm)
                for s in solvers:
                    s.backend.unpack(field, s.recv, s.neighbours)
            for s in solvers:
                s.backend.phase(phase, 2e-7)
    got = np.empty_like(want)
    for s in solvers:
        si, sj = s.grid.tile_slices(nx, ny)
        got[si, sj] = s.local_state()
    np.testing.assert_array_equal(got, want)


def test_generic_kernel_in_padded_layout():
    """tiles the LDS kernels do not cover (24 x 40) go through the generic kernel, halo reads included"""
    rng = np.random.default_rng(5)
    px, py, tx, ty = 2, 2, 24, 40
    nx, ny = px * tx, py * ty
    dom = std_domain(P, nx, ny)
    eq = P.AllenCahn2DPeriodic(dom, 0.002, MU["cubic"], MOB["one_plus_sq"])
    y0 = 0.1 * rng.standard_normal((nx, ny))
    eng = P.HipEngine()
    eng.set_kernel_path(L.PATH_GENERIC)  # same kernel as the padded tiles -> bitwise comparable
    want = P.diffeqsolve(eq, P.RK4(), 0.0, 3 * 1e-4, 1e-4, y0, engine=eng).ys[-1]
    assert "generic" in eng.last_kernel
    eng.close()
    comm = InProcessComm(px * py)
    solvers = []
    for r in range(px * py):
        be = HipTileBackend(eq, (tx, ty), np.float64)
        s = DecomposedSolver(eq, CartesianGrid(px, py, r), comm=comm.view(r), dtype=np.float64, backend=be)
        s.set_global_state(y0)
        solvers.append(s)
    plan = solvers[0].backend.phase_plan()
    assert len(plan) == 4
    for _ in range(3):
        for phase, field in enumerate(plan):
            for s in solvers:
                s.backend.pack(field, s.send)
            gather_all(comm)
            for s in solvers:
                s.backend.unpack(field, s.recv, s.neighbours)
                s.backend.phase(phase, 1e-4)
    got = np.empty_like(want)
    for s in solvers:
        si, sj = s.grid.tile_slices(nx, ny)
        got[si, sj] = s.local_state()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(64, 128), (96, 384), (256, 256)])
def test_interior_plus_edge_launches_equal_one_launch(dtype, shape):
    """pdeopt_rk4_phase_part: the interior tiles (run while the halo exchange is in flight) and the edge tiles
    (after the unpack) together do exactly what the single launch does -- bitwise, also when a tile row /
    column is both first and last (64 x 128: every tile is an edge tile)"""
    rng = np.random.default_rng(6)
    nx, ny = shape
    dom = std_domain(P, nx, ny)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(dtype)
    outs = []
    for split in (False, True):
        be = HipTileBackend(eq, (nx, ny), dtype, halo=4)
        be.set_state(y0)
        plan = be.phase_plan()
        assert len(plan) == 2
        for _ in range(5):
            for phase, field in enumerate(plan):
                be.pack(field, None)
                if split:
                    be.phase(phase, 2e-7, L.PART_INTERIOR)  # reads no halo cell: may precede the unpack
                    be.unpack(field, None, [0] * 8)
                    be.phase(phase, 2e-7, L.PART_EDGE)
                else:
                    be.unpack(field, None, [0] * 8)
                    be.phase(phase, 2e-7)
        outs.append(be.get_state())
        be.engine.close()
    np.testing.assert_array_equal(outs[1], outs[0])
    want, _ = _monolithic(eq, y0, 2e-7, 5, 0)
    np.testing.assert_array_equal(outs[0], want)


@pytest.mark.parametrize("grid,tile,halo,fuse", [((2, 2), (64, 128), 8, 0), ((2, 2), (64, 128), 8, 1), ((2, 2), (64, 128), 4, 0),
                                                 ((2, 2), (64, 128), 4, -1), ((1, 2), (96, 256), 8, 0), ((4, 1), (32, 128), 8, 0),
                                                 ((2, 3), (64, 128), 8, 0), ((2, 2), (64, 64), 8, 0), ((1, 2), (96, 256), 8, 1)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_in_library_loop_with_virtual_ranks_equals_monolithic(grid, tile, halo, fuse, dtype):
    """pdeopt_rk4_decomposed_advance -- the substep loop in C with its neighbour table and rank offsets into the
    gathered strips -- on px x py VIRTUAL ranks: engines of this process on one GPU joined in an in-process group
    (pdeopt_comm_init_local), one host thread per rank, the all-gather = event-ordered device copies between the
    ranks' strip buffers.  Everything the RCCL run executes except the collective itself; bitwise equal to the
    monolithic periodic solve, also across two advance calls (the exchange parity continues) and on grids where
    a rank is its own neighbour (1 x 2, 4 x 1)."""
    px, py = grid
    tx, ty = tile
    if dtype is np.float64 and tile == (64, 64):
        pytest.skip("64 x 64 tiles belong to the fp32 whole-substep kernel (fp64 runs the stage pairs: 32-vector rows)")
    if dtype is np.float64:
        ty //= 2  # 16-byte vectors hold 2 cells: the 32-vector tile is 64 columns wide
    nx, ny = px * tx, py * ty
    rng = np.random.default_rng(7)
    dom = std_domain(P, nx, ny)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(dtype)
    want, _ = _monolithic(eq, y0, 2e-7, 7, fuse)
    comms = LocalGroupComm.create(px * py)
    solvers = []
    for r in range(px * py):
        be = HipTileBackend(eq, (tx, ty), dtype, halo=halo)
        be.engine.set_fuse_stages(fuse)
        s = DecomposedSolver(eq, CartesianGrid(px, py, r), comm=comms[r], dtype=dtype, backend=be)
        s.set_global_state(y0)
        solvers.append(s)
    advance_group(solvers, 2e-7, 4)
    advance_group(solvers, 2e-7, 3)
    got = np.empty_like(want)
    for s in solvers:
        assert s.mode == "local-group"
        # halo 8, fp32, tiles of 32 x 128 or 64 x 64 cells: ONE kernel per substep (fused unpack in, fused pack out)
        quad = halo == 8 and fuse == 0 and dtype is np.float32
        assert ("rk4_quad<f32,CH,halo8" in s.backend.engine.last_kernel) == quad, s.backend.engine.last_kernel
        si, sj = s.grid.tile_slices(nx, ny)
        got[si, sj] = s.local_state()
    np.testing.assert_array_equal(got, want)
    for s in solvers:
        s.backend.engine.close()
    comms[0].group.close()


def _run_peer_mapped(px, py, tile, dtype, calls, sabotage=False):
    import multiprocessing as mp
    import threading

    from decomp_util import peer_mapped_worker

    world = px * py
    nx, ny = px * tile[0], py * tile[1]
    rng = np.random.default_rng(21)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(dtype)
    ctx = mp.get_context("spawn")
    q_up, q_out = ctx.Queue(), ctx.Queue()
    q_down = [ctx.Queue() for _ in range(world)]
    procs = [ctx.Process(target=peer_mapped_worker, args=(r, px, py, (nx, ny), y0, 2e-7, calls, np.dtype(dtype).name, q_up, q_down[r], q_out,
                                                          sabotage)) for r in range(world)]
    for p_ in procs:
        p_.start()

    def gather():  # the "transport" of the handles: every rank's 64 bytes to every rank
        got = dict(q_up.get(timeout=180) for _ in range(world))
        for r in range(world):
            q_down[r].put([got[i] for i in range(world)])

    t = threading.Thread(target=gather)
    t.start()
    try:
        results = dict((r, (mode, kern, tile_)) for r, mode, kern, tile_ in (q_out.get(timeout=240) for _ in range(world)))
    finally:
        t.join(timeout=10)
        for r in range(world):
            q_down[r].put("done")
        for p_ in procs:
            p_.join(timeout=60)
            if p_.is_alive():
                p_.kill()
    return y0, (nx, ny), results


@pytest.mark.parametrize("grid,tile,dtype", [((2, 1), (64, 128), np.float32), ((2, 2), (64, 128), np.float32), ((2, 2), (64, 64), np.float64)])
def test_peer_mapped_exchange_between_processes_sharing_one_gpu(grid, tile, dtype):
    """pdeopt_comm_ipc_export / _attach ("P2P stores into peer-mapped halo buffers"): one PROCESS per rank
    as in the process-per-GPU deployment, here all on GPU 0 (hipIpc handles work across processes on one device).  Every
    rank's stencil kernel reads its neighbours' strips in place through the mapped buffers and writes its own; counters
    order the exchanges; no collective.  Bitwise equal to the monolithic periodic solve, across two advance calls."""
    px, py = grid
    y0, (nx, ny), res = _run_peer_mapped(px, py, tile, dtype, calls=(4, 3))
    assert all(v[0] == "peer-mapped" for v in res.values()), {r: v[:2] for r, v in res.items()}
    quad = dtype is np.float32
    assert all(("rk4_quad<f32,CH,halo8" in v[1]) == quad for v in res.values()), [v[1] for v in res.values()]
    dom = std_domain(P, nx, ny)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    want, _ = _monolithic(eq, y0, 2e-7, 7, 0)
    got = np.empty_like(want)
    for r, (_, _, t_) in res.items():
        si, sj = CartesianGrid(px, py, r).tile_slices(nx, ny)
        got[si, sj] = t_
    np.testing.assert_array_equal(got, want)


def test_peer_mapped_exchange_absent_rank_is_an_error_not_a_hang():
    """a neighbour that never publishes its strip: the wait kernel gives up after 2 s and the call fails on the ranks
    that waited (no wave spins forever)"""
    _, _, res = _run_peer_mapped(2, 1, (64, 128), np.float32, calls=(3,), sabotage=True)
    assert res[1][0] == "absent"
    assert res[0][0] == "error" and "did not publish" in res[0][1], res[0]


@pytest.mark.parametrize("grid,tile,dtype", [((2, 2), (64, 128), np.float32), ((1, 2), (96, 256), np.float32), ((2, 2), (64, 128), np.float64)])
def test_virtual_ranks_gathered_copy_path_equals_monolithic(grid, tile, dtype, monkeypatch):
    """Ranks of one process on ONE device read each other's strips in place (no copies: the default, covered above); the
    event-ordered device copies into a gathered buffer are what in-process ranks on DIFFERENT devices use.  One GPU here:
    PDEOPT_LOCAL_GATHER keeps that path selectable -- bitwise the same result."""
    monkeypatch.setenv("PDEOPT_LOCAL_GATHER", "1")
    px, py = grid
    tx, ty = tile
    if dtype is np.float64:
        ty //= 2
    nx, ny = px * tx, py * ty
    dom = std_domain(P, nx, ny)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.05 * np.random.default_rng(11).standard_normal((nx, ny)), 0.05, 0.95).astype(dtype)
    want, _ = _monolithic(eq, y0, 2e-7, 7, 0)
    comms = LocalGroupComm.create(px * py)
    solvers = []
    for r in range(px * py):
        s = DecomposedSolver(eq, CartesianGrid(px, py, r), comm=comms[r], dtype=dtype, backend=HipTileBackend(eq, (tx, ty), dtype, halo=8))
        s.set_global_state(y0)
        solvers.append(s)
    advance_group(solvers, 2e-7, 4)
    advance_group(solvers, 2e-7, 3)
    got = np.empty_like(want)
    for s in solvers:
        si, sj = s.grid.tile_slices(nx, ny)
        got[si, sj] = s.local_state()
    np.testing.assert_array_equal(got, want)
    for s in solvers:
        s.backend.engine.close()
    comms[0].group.close()


def test_local_group_misuse_is_an_error_not_a_hang():
    """a rank whose partners never arrive fails with a message (the rendezvous has a time-out; here a rank count
    mismatch is detected before any waiting)"""
    dom = std_domain(P, 64, 128)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    comms = LocalGroupComm.create(2)
    s0 = DecomposedSolver(eq, CartesianGrid(2, 1, 0), comm=comms[0], dtype=np.float32)
    with pytest.raises(ValueError, match="taken"):
        DecomposedSolver(eq, CartesianGrid(2, 1, 0), comm=LocalGroupComm(comms[0].group, 0), dtype=np.float32)
    with pytest.raises(ValueError, match="communicator has 2 ranks"):
        DecomposedSolver(eq, CartesianGrid(1, 1, 0), comm=comms[1], dtype=np.float32)
    s0.backend.engine.close()
    comms[0].group.close()


def test_config5_full_size_virtual_ranks_vs_c_oracle():
    """BASELINE config 5 at its real size: the 4096^2 field on 2 x 2 ranks of 2048^2 -- virtual ranks of this process
    on one GPU running the library's decomposed loop (one whole-substep kernel + one exchange per substep) -- ALL FOUR
    tiles against oracle/c_oracle.c on the whole periodic field (cahn_hilliard.py:89-109), and bitwise against the
    monolithic GPU solve"""
    from oracle import c_oracle as CO

    n, nsub, dt = 4096, 7, 2e-7
    rng = np.random.default_rng(0)
    dom = std_domain(P, n, n)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((n, n)), 0.05, 0.95).astype(np.float32)
    comms = LocalGroupComm.create(4)
    solvers = []
    for r in range(4):
        s = DecomposedSolver(eq, CartesianGrid(2, 2, r), comm=comms[r], dtype=np.float32)
        s.set_global_state(y0)
        solvers.append(s)
    advance_group(solvers, dt, 4)
    advance_group(solvers, dt, nsub - 4)
    got = np.empty_like(y0)
    for s in solvers:
        assert "rk4_quad<f32,CH,halo8" in s.backend.engine.last_kernel, s.backend.engine.last_kernel
        si, sj = s.grid.tile_slices(n, n)
        got[si, sj] = s.local_state()
        s.backend.engine.close()
    comms[0].group.close()
    hx, hy = dom.dx
    regsol = CO.closure(0, 1, (3.0, -6.0))  # polynomial 3 - 6 c + the logit flag
    c1mc = CO.closure(0, 0, (0.0, 1.0, -1.0))
    ref = CO.rk4(0, y0, hx, hy, 0.002, regsol, c1mc, dt, nsub)
    inc, inc_ref = got.astype(np.float64) - y0, ref.astype(np.float64) - y0
    for r in range(4):  # every rank's tile on its own
        si, sj = CartesianGrid(2, 2, r).tile_slices(n, n)
        assert np.max(np.abs(got[si, sj] - ref[si, sj])) < 5e-7, (r, float(np.max(np.abs(got[si, sj] - ref[si, sj]))))
        e = np.linalg.norm(inc[si, sj] - inc_ref[si, sj]) / np.linalg.norm(inc_ref[si, sj])
        assert e < 5e-5, (r, e)
    want, kern = _monolithic(eq, y0, dt, nsub, 0)
    assert "rk4_quad" in kern
    np.testing.assert_array_equal(got, want)


def test_config5_tile_size_smoke():
    """one 2048^2 tile of BASELINE config 5 (4096^2 over 2x2) with self-neighbours: finite + mass"""
    rng = np.random.default_rng(2)
    dom = std_domain(P, 2048, 2048)
    eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((2048, 2048)), 0.05, 0.95).astype(np.float32)
    s = DecomposedSolver(eq, CartesianGrid(1, 1, 0), dtype=np.float32)
    s.set_global_state(y0)
    s.advance(2e-7, 10)
    y1 = s.local_state()
    assert np.all(np.isfinite(y1))
    assert abs(y1.astype(np.float64).mean() - y0.astype(np.float64).mean()) < 2e-7


def test_rccl_allgather_path_single_rank():
    """the production exchange path: strips in torch device tensors, all-gathered by RCCL
    (torch.distributed backend 'nccl'), engine work ordered on torch's current stream"""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from pde_opt_amd.decomp import TorchComm

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # single node: no interface probing
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(4)
        dom = std_domain(P, 64, 128)
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        y0 = np.clip(0.5 + 0.05 * rng.standard_normal((64, 128)), 0.05, 0.95).astype(np.float32)
        comm = TorchComm()
        assert comm.stream_handle  # a real (non-null) stream shared by RCCL and the engine
        want, _ = _monolithic(eq, y0, 2e-7, 51, 0)
        modes, errors = {}, []
        for graph, overlap in ((True, True), (False, True), (False, False)):
            sol = DecomposedSolver(eq, CartesianGrid(1, 1, 0), comm=comm, dtype=np.float32, halo=4)
            sol.use_graph, sol.use_overlap = graph, overlap
            sol.set_global_state(y0)
            sol.advance(2e-7, 51)  # odd: graph-replayed pairs + one eager substep
            torch.cuda.synchronize()
            got = sol.local_state()
            bad = np.abs(got - want)
            print("mode", sol.mode, "max diff", float(bad.max()), "rows", np.where(bad.max(axis=1) > 0)[0][:8], flush=True)
            if bad.max() > 0:
                errors.append((graph, overlap, sol.mode, float(bad.max())))
            assert sol.send.is_cuda and sol.recv.numel() == sol.backend.strip_elems
            assert sol.exchanges == 2 * 51
            modes[(graph, overlap)] = sol.mode
            sol.backend.engine.close()
            if graph and sol.mode != "graph+overlap":
                print("device-graph capture of the exchange unavailable:", getattr(sol, "capture_error", None))
        # the library's own RCCL communicator: the whole loop in C, with and without the overlapped collective
        from pde_opt_amd.decomp import NativeComm

        for overlap, halo in ((True, 4), (False, 4), (False, 8)):
            sol = DecomposedSolver(eq, CartesianGrid(1, 1, 0), comm=NativeComm(), dtype=np.float32, halo=halo)
            sol.use_overlap = overlap
            sol.set_global_state(y0)
            sol.advance(2e-7, 30)
            sol.advance(2e-7, 21)
            got = sol.local_state()
            print("mode", sol.mode, "halo", halo, "max diff", float(np.abs(got - want).max()), flush=True)
            assert sol.mode == ("native+overlap" if overlap else "native")
            assert sol.exchanges == (2 if halo == 4 else 1) * 51
            if not np.array_equal(got, want):
                errors.append(("native", overlap, halo, sol.mode, float(np.abs(got - want).max())))
            sol.backend.engine.close()
        # the plain torch all-gather driver on the halo-8 layout (one exchange per substep, pack / unpack kernels)
        sol = DecomposedSolver(eq, CartesianGrid(1, 1, 0), comm=comm, dtype=np.float32, halo=8)
        sol.set_global_state(y0)
        sol.advance(2e-7, 51)
        torch.cuda.synchronize()
        if not np.array_equal(sol.local_state(), want):
            errors.append(("torch plain", 8, float(np.abs(sol.local_state() - want).max())))
        assert sol.mode == "plain" and sol.exchanges == 51
        sol.backend.engine.close()
        assert not errors, errors
        assert modes[(False, True)] == "overlap" and modes[(False, False)] == "plain"
        assert modes[(True, True)] in ("graph+overlap", "overlap")
        print("decomposed driver modes:", modes)
    finally:
        dist.destroy_process_group()
