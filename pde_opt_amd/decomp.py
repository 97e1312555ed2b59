"""Domain decomposition of ONE large periodic field over a px x py grid of ranks (BASELINE config 5:
Cahn-Hilliard 4096^2 on 2x2 GPUs).  New relative to the reference, which has no multi-device code
(SURVEY 2.1); semantics = the monolithic periodic solve, bit for bit for the explicit integrators.

Every rank owns an (nx/px) x (ny/py) tile stored with a halo.  Per RK4 substep::

    for phase, field in enumerate(backend.phase_plan()):       # 2 phases (fused pairs) or 4
        if field >= 0:                                         # halo 8: only phase 0 exchanges (the state)
            backend.pack(field, send)                          # 8 interior pieces -> one strip
            comm.all_gather(send, recv)                        # RCCL over xGMI (torch.distributed)
            backend.unpack(field, recv, grid.neighbours())     # 8 halo pieces from 8 neighbours
        backend.phase(phase, dt)                               # fused stencil + RK update kernel

Two layouts.  **Halo 8** (default where the fused Cahn-Hilliard kernels run): ONE exchange per substep.  fp32 tiles that
32 x 128 (or 64 x 64) cells divide run the WHOLE substep in one kernel (``csrc/stencil_fused_ch4.hpp``: its tile + 8
input is exactly the 8-cell halo; plan ``[0]``): in the library's own loops the edge workgroups read the halo straight
from the gathered strips and write the new state's strip themselves -- a substep is 1 stencil kernel + 1 collective.
Otherwise (fp64, other tile shapes, ``PDEOPT_OPT_FUSE_STAGES = 1``; plan ``[0, -1]``) the first stage pair is evaluated on
the tile + 4 ring (it reads the state on tile + 8), so the second pair finds its input there: 2 stencil kernels + 1
collective.  **Halo 4**: one exchange per phase (2 per
substep with fused pairs, 4 with per-stage kernels), the layout of the interior / edge overlap and graph drivers.

The exchange is ONE all-gather of packed strips per phase (edges and corners together), the pattern
BASELINE.json names; strips are 2*h*(nx+ny)+4*h^2 elements (128 KiB at 2048^2 fp32), far below the
per-link xGMI bandwidth, so the collective is latency-bound: fewer, fatter exchanges (halo 4 for a
fused stage pair) is one lever, hiding the latency the other.  On the GPU the driver therefore

  * can overlap (``use_overlap``): the all-gather runs on its own stream while the INTERIOR workgroup tiles of
    the phase (the ones that read no halo cell, ~85 % of a 2048^2 tile) are computed; the EDGE tiles follow
    the unpack (``pdeopt_rk4_phase_part``);
  * can replay (``use_graph``, TorchComm): two substeps (pack, collective, interior, unpack, edge x 2 phases
    x 2) are captured once into a device graph (``torch.cuda.graph`` on the engine's stream, the RCCL
    collective included) and replayed, so a substep costs the host half a graph launch instead of ~18 ctypes /
    torch calls;
  * a single rank (loop-back exchange) runs the whole substep loop inside the library
    (``pdeopt_rk4_loopback_advance``).

  * ``NativeComm``: the library opens its own RCCL communicator (``pdeopt_comm_init``, the id broadcast by the
    host) and runs the WHOLE loop -- pack, ``ncclAllGather``, interior tiles on the compute stream while the
    collective is in flight on a second stream, unpack, edge tiles -- inside ``pdeopt_rk4_decomposed_advance``:
    no Python, no graph, ~10 asynchronous launches per substep issued from C.

``DecomposedSolver.mode`` reports which path ran ("native+overlap", "native", "loopback", "graph+overlap",
"overlap", "plain"); a failed capture falls back to the eager overlapped path, CPU backends (gloo tests) to the
plain sequence above.  Measured on ONE rank (a 1-rank RCCL group, 2048^2 tile, us per substep): in-library
loop-back 46; torch all-gather plain 69 (host-bound: ~8 calls per substep), eager overlap 111 (more host
calls), graph replay 99 (HIP graph launches cost ~5 us per node); see DESIGN.md section 6 for the native path.
IMEX / Strang would need a distributed FFT (all-to-all transposes): replicas only for the spectral
integrators.

``TileBackend`` is the HIP engine by default; tests inject a CPU backend built on the oracle to
check the exchange protocol under gloo (world_size 2 and 4) without a GPU.
"""

from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L
from .engine import HipEngine

HALO = 4  # halo width of the per-phase exchange layout (the oracle tile backend of the CPU tests uses it)


class CartesianGrid:
    """px x py periodic process grid, rank = ri * py + rj (row-major; ri indexes x = axis 0)."""

    def __init__(self, px: int, py: int, rank: int):
        if not 0 <= rank < px * py:
            raise ValueError(f"rank {rank} outside a {px}x{py} grid")
        self.px, self.py, self.rank = px, py, rank
        self.ri, self.rj = divmod(rank, py)

    @property
    def world(self) -> int:
        return self.px * self.py

    def rank_of(self, ri: int, rj: int) -> int:
        return (ri % self.px) * self.py + (rj % self.py)

    def neighbours(self) -> List[int]:
        """[up, down, left, right, up-left, up-right, down-left, down-right]; up = smaller x index."""
        i, j = self.ri, self.rj
        return [
            self.rank_of(i - 1, j), self.rank_of(i + 1, j), self.rank_of(i, j - 1), self.rank_of(i, j + 1),
            self.rank_of(i - 1, j - 1), self.rank_of(i - 1, j + 1), self.rank_of(i + 1, j - 1), self.rank_of(i + 1, j + 1),
        ]

    def tile_shape(self, nx: int, ny: int):
        if nx % self.px or ny % self.py:
            raise ValueError(f"grid {nx}x{ny} does not divide over {self.px}x{self.py} ranks")
        return nx // self.px, ny // self.py

    def tile_slices(self, nx: int, ny: int):
        tx, ty = self.tile_shape(nx, ny)
        return slice(self.ri * tx, (self.ri + 1) * tx), slice(self.rj * ty, (self.rj + 1) * ty)


def strip_layout(nx: int, ny: int, h: int = HALO):
    """(offset, shape) of the 8 pieces of a strip, in kernel order (csrc/halo.hip):
    top rows, bottom rows, left cols, right cols, TL, TR, BL, BR."""
    shapes = [(h, ny), (h, ny), (nx, h), (nx, h), (h, h), (h, h), (h, h), (h, h)]
    out, off = [], 0
    for s in shapes:
        out.append((off, s))
        off += s[0] * s[1]
    return out, off


# piece q of MY halo comes from piece FROM[q] of neighbour q
FROM = (1, 0, 3, 2, 7, 6, 5, 4)


# ---------------------------------------------------------------------------------- communicators
class LoopbackComm:
    """world_size 1: the library's internal device buffer stands in for the collective."""

    world, rank = 1, 0

    def make_buffers(self, backend):
        return None, None

    def all_gather(self, send, recv):
        pass


class TorchComm:
    """torch.distributed all-gather of strips: backend 'nccl' (= RCCL over xGMI) on device tensors,
    'gloo' on CPU tensors (tests)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist

        self._dist = dist
        self._torch = torch
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # On the GPU the collectives are issued under a dedicated torch stream and the HIP engine
        # runs on the SAME stream (stream_handle), so pack -> all-gather -> unpack -> stencil order
        # without host synchronisation.  (torch's default stream is the null stream, whose handle is
        # 0 and cannot be shared with a non-blocking stream: an explicit stream is required.)
        self.stream = None
        self.stream_handle = None
        self.comm_stream = None
        if dist.get_backend(group) == "nccl":
            self.stream = torch.cuda.Stream()         # the engine's stream: kernels, pack / unpack
            self.stream_handle = self.stream.cuda_stream
            self.comm_stream = torch.cuda.Stream()    # collectives, overlapped with the interior tiles
            self.ev_packed = torch.cuda.Event()
            self.ev_gathered = torch.cuda.Event()

    def make_buffers(self, backend):
        import torch

        dt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(backend.dtype)]
        dev = torch.device("cuda", backend.device) if backend.on_device else torch.device("cpu")
        send = torch.empty(backend.strip_elems, dtype=dt, device=dev)
        recv = torch.empty(self.world * backend.strip_elems, dtype=dt, device=dev)
        return send, recv

    def all_gather(self, send, recv):
        if self.stream is not None:
            with self._torch.cuda.stream(self.stream):
                self._dist.all_gather_into_tensor(recv, send, group=self.group)
        else:
            self._dist.all_gather_into_tensor(recv, send, group=self.group)

    @property
    def can_overlap(self) -> bool:
        return self.comm_stream is not None

    def all_gather_overlapped(self, send, recv):
        """the collective on the communication stream, ordered after everything issued so far on the engine's
        stream (the pack); ``wait_gathered`` makes the engine's stream wait for it"""
        self.ev_packed.record(self.stream)
        self.comm_stream.wait_event(self.ev_packed)
        with self._torch.cuda.stream(self.comm_stream):
            self._dist.all_gather_into_tensor(recv, send, group=self.group)
        self.ev_gathered.record(self.comm_stream)

    def wait_gathered(self):
        self.stream.wait_event(self.ev_gathered)

    def capture(self, body):
        """record ``body()`` (work on the engine's stream + the overlapped collectives) into a device graph"""
        torch = self._torch
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            body()
        return g

    def replay(self, g):
        """launch a captured graph ON THE ENGINE'S STREAM (``g.replay()`` alone would use torch's current
        stream, unordered with the engine's eager work before and after it)"""
        with self._torch.cuda.stream(self.stream):
            g.replay()


class NativeComm:
    """The library's own RCCL communicator (csrc/comm.hip): the whole substep loop -- pack, ncclAllGather of
    strips, unpack, stencil phases, the collective overlapped with the interior tiles on a second HIP stream --
    runs inside ``pdeopt_rk4_decomposed_advance`` with no Python per substep.  The host side only carries rank
    0's 128-byte RCCL id to the other ranks: ``broadcast(id_or_None) -> id`` (default: torch.distributed's
    object broadcast, whatever its backend)."""

    def __init__(self, world: Optional[int] = None, rank: Optional[int] = None, broadcast=None):
        if world is None or rank is None:
            import torch.distributed as dist

            world, rank = dist.get_world_size(), dist.get_rank()
        self.world, self.rank = int(world), int(rank)
        self._broadcast = broadcast or self._torch_broadcast
        self._attached = False

    @staticmethod
    def _torch_broadcast(uid):
        import torch.distributed as dist

        box = [uid]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def make_buffers(self, backend):
        if not self._attached:
            eng = backend.engine
            uid = self._broadcast(eng.comm_unique_id() if self.rank == 0 else None)
            eng.comm_init(self.world, self.rank, uid)  # collective: every rank calls it
            self._attached = True
        return None, None  # the strip buffers live in the library

    def all_gather(self, send, recv):
        raise RuntimeError("NativeComm exchanges inside pdeopt_rk4_decomposed_advance")


class PeerMappedComm:
    """One process per GPU, NO collective in the substep (csrc/comm.hip, peer-mapped exchange): every
    rank's strip buffers are mapped into every process (hipIpc), a rank's stencil kernel reads its neighbours' strips in
    place and writes its own next strip, two counters per rank order the exchanges.  ``allgather(obj) -> [obj of rank 0,
    ..., obj of rank world-1]`` carries the 64-byte handles between the processes once, at set-up (default:
    ``torch.distributed.all_gather_object``); processes sharing one GPU work too."""

    def __init__(self, world: Optional[int] = None, rank: Optional[int] = None, allgather=None):
        if world is None or rank is None:
            import torch.distributed as dist

            world, rank = dist.get_world_size(), dist.get_rank()
        self.world, self.rank = int(world), int(rank)
        self._allgather = allgather or self._torch_allgather
        self._attached = False

    def _torch_allgather(self, obj):
        import torch.distributed as dist

        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def make_buffers(self, backend):
        if not self._attached:
            if getattr(backend, "halo", 4) != 8:
                raise ValueError("the peer-mapped exchange needs the halo-8 layout (fused Cahn-Hilliard kernels)")
            eng = backend.engine
            handles = self._allgather(eng.comm_ipc_export(self.world, self.rank))
            eng.comm_ipc_attach(handles)
            self._attached = True
        return None, None  # the strip buffers live in the library

    def all_gather(self, send, recv):
        raise RuntimeError("PeerMappedComm exchanges inside pdeopt_rk4_decomposed_advance")


class LocalGroupComm:
    """One rank of an IN-PROCESS group (``pdeopt_local_group``, csrc/comm.hip): the ranks are engines of this
    process -- several on one GPU ("virtual ranks": the library's decomposed loop, its neighbour tables and rank
    offsets run with world > 1 on a single GPU) or one per GPU -- and the all-gather is device-side copies between
    their strip buffers ordered by HIP events, inside the same ``pdeopt_rk4_decomposed_advance`` loop the RCCL
    communicator runs.  Every rank's ``DecomposedSolver.advance`` must be called at the same time from its own
    host thread: ``advance_group(solvers, dt, n)`` does that."""

    def __init__(self, group, rank: int):
        self.group, self.world, self.rank = group, group.world, int(rank)
        self._attached = False

    @staticmethod
    def create(world: int):
        """[LocalGroupComm(rank) for rank in range(world)] sharing one group"""
        from .engine import LocalGroup

        g = LocalGroup(world)
        return [LocalGroupComm(g, r) for r in range(world)]

    def make_buffers(self, backend):
        if not self._attached:
            backend.engine.comm_init_local(self.group, self.rank)
            self._attached = True
        return None, None  # the strip buffers live in the library

    def all_gather(self, send, recv):
        raise RuntimeError("LocalGroupComm exchanges inside pdeopt_rk4_decomposed_advance")


def advance_group(solvers, dt: float, n_substeps: int):
    """``advance`` on every rank of an in-process group at the same time, one host thread per rank (the ranks
    rendezvous once per halo exchange inside the library); the first failure is re-raised after all have returned"""
    import threading

    errors = [None] * len(solvers)

    def run(i):
        try:
            solvers[i].advance(dt, n_substeps)
        except BaseException as e:  # noqa: BLE001
            errors[i] = e

    threads = [threading.Thread(target=run, args=(i,), name=f"pdeopt-rank{i}") for i in range(len(solvers))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e


# ---------------------------------------------------------------------------------- HIP backend
class HipTileBackend:
    """One rank's tile on the GPU (padded layout of libpdeopt_hip.so)."""

    on_device = True

    def __init__(self, equation, tile_shape, dtype=np.float32, device: int = 0, stream: Optional[int] = None,
                 engine: Optional[HipEngine] = None, halo: Optional[int] = None):
        """``halo``: 8 (one exchange per substep; fused Cahn-Hilliard stage pairs), 4 (one per phase), or None =
        8 where the library runs the fused pairs on this problem, else 4"""
        self.engine = engine or HipEngine(device, stream=stream)
        self.device = device
        self.dtype = np.dtype(dtype)
        prob = dict(equation._engine_problem())
        prob["nx"], prob["ny"] = tile_shape  # same spacing, local extent
        if halo not in (None, 4, 8):
            raise ValueError(f"halo must be 4 or 8, got {halo}")
        for h in ((8, 4) if halo is None else (halo,)):
            self.engine.set_halo_layout(h)
            self.engine.configure(dtype=dtype, batch=1, **prob)
            self.engine.set_halo_layout(0)  # the option only applies to the configure above
            self.halo = h
            # [0]: the whole substep in one kernel (fp32, tiles of 32 x 128 / 64 x 64 cells); [0, -1]: two stage pairs
            if h != 8 or self.engine.rk4_phase_plan() in ([0], [0, -1]):
                break
            if halo == 8:
                raise ValueError("halo=8 needs the fused Cahn-Hilliard kernels (closure class, tile shape and "
                                 "PDEOPT_OPT_FUSE_STAGES decide); this problem runs one kernel per stage: use halo=4")
        self.strip_elems = self.engine.halo_strip_elems()

    def phase_plan(self) -> List[int]:
        return self.engine.rk4_phase_plan()

    @staticmethod
    def _ptr(buf):
        return None if buf is None else int(buf.data_ptr())

    def pack(self, field: int, send):
        self.engine.halo_pack(field, self._ptr(send))

    def unpack(self, field: int, recv, neighbours: Sequence[int]):
        self.engine.halo_unpack(field, self._ptr(recv), neighbours)

    def phase(self, phase: int, dt: float, part: int = 0):
        self.engine.rk4_phase(phase, dt, part)

    def loopback_advance(self, dt: float, n: int):
        self.engine.rk4_loopback_advance(dt, n)

    def set_state(self, tile):
        self.engine.set_state(np.asarray(tile, dtype=self.dtype))

    def get_state(self):
        return self.engine.get_state()[0]


# ---------------------------------------------------------------------------------- driver
class DecomposedSolver:
    """RK4 on one rank's tile of a decomposed periodic field."""

    def __init__(self, equation, grid: CartesianGrid, comm=None, dtype=np.float32, device: int = 0,
                 backend=None, stream: Optional[int] = None, halo: Optional[int] = None):
        self.equation = equation
        self.grid = grid
        self.comm = comm or LoopbackComm()
        if self.comm.world != grid.world:
            raise ValueError(f"communicator has {self.comm.world} ranks, process grid {grid.world}")
        nx, ny = equation.domain.points
        self.tile_shape = grid.tile_shape(nx, ny)
        if min(self.tile_shape) < 16:
            raise ValueError("tiles must be at least 16 cells wide")
        if stream is None:
            stream = getattr(self.comm, "stream_handle", None)
        self.backend = backend or HipTileBackend(equation, self.tile_shape, dtype, device, stream, halo=halo)
        self.send, self.recv = self.comm.make_buffers(self.backend)
        self.neighbours = grid.neighbours()
        self.exchanges = 0
        self.mode = "plain"
        # Both default OFF, by measurement on this stack (one rank, us per substep of a 2048^2 / 4096^2 tile):
        # in-library loop with the collective on the compute stream 51 / 116; the same with the collective on a
        # second stream 114 / 187 -- the two cross-stream event hops of a phase cost ~30 us, more than the interior
        # tiles (20 / 50 us) can hide unless the collective itself takes longer than that; torch all-gather plain
        # 74 / 134, eager overlap 111 / 170, graph replay 99 / 179 (HIP graph launches: ~5 us per node).
        self.use_graph = False    # device-graph replay of substep pairs (TorchComm)
        self.use_overlap = False  # interior tiles overlap the collective (fused stage pairs on the GPU)
        self._graph = None
        self._graph_dt = None

    def set_global_state(self, u_global):
        si, sj = self.grid.tile_slices(*self.equation.domain.points)
        self.backend.set_state(np.asarray(u_global)[si, sj])

    def exchange(self, field: int):
        if field < 0:  # halo-8 layout: this phase finds its input's ring computed by the previous one
            return
        self.backend.pack(field, self.send)
        self.comm.all_gather(self.send, self.recv)
        self.backend.unpack(field, self.recv, self.neighbours)
        self.exchanges += 1

    def _substep_overlapped(self, plan, dt):
        be, c = self.backend, self.comm
        for phase, field in enumerate(plan):
            be.pack(field, self.send)
            c.all_gather_overlapped(self.send, self.recv)     # on the communication stream ...
            be.phase(phase, dt, L.PART_INTERIOR)              # ... while the interior tiles run
            c.wait_gathered()
            be.unpack(field, self.recv, self.neighbours)
            be.phase(phase, dt, L.PART_EDGE)

    def advance(self, dt: float, n_substeps: int):
        n = int(n_substeps)
        plan = self.backend.phase_plan()
        be, c = self.backend, self.comm
        nex = sum(1 for f in plan if f >= 0)  # exchanges per substep
        if (self.use_overlap or self.use_graph) and getattr(be, "halo", HALO) == 8:
            raise ValueError("the interior / edge overlap and the graph replay belong to the halo-4 layout "
                             "(DecomposedSolver(..., halo=4)); halo 8 exchanges once per substep")
        if isinstance(c, (NativeComm, LocalGroupComm, PeerMappedComm)):
            self.mode = "native" if isinstance(c, NativeComm) else ("peer-mapped" if isinstance(c, PeerMappedComm) else "local-group")
            if self.use_overlap and len(plan) == 2 and isinstance(c, NativeComm):
                self.mode = "native+overlap"
            be.engine.rk4_decomposed_advance(dt, n, self.neighbours, overlap=self.use_overlap)
            self.exchanges += n * nex
            return
        if isinstance(c, LoopbackComm) and hasattr(be, "loopback_advance"):
            self.mode = "loopback"
            be.loopback_advance(dt, n)  # the whole loop inside the library
            self.exchanges += n * nex
            return
        overlap = self.use_overlap and getattr(c, "can_overlap", False) and getattr(be, "on_device", False) and len(plan) == 2
        if not overlap:
            self.mode = "plain"
            for _ in range(n):
                for phase, field in enumerate(plan):
                    self.exchange(field)
                    be.phase(phase, dt)
            return
        done = 0
        if self.use_graph and n >= 4:
            if self._graph is None or self._graph_dt != dt:
                # buffers, the communicator and the kernels' lazy allocations exist before the capture
                for _ in range(2):
                    self._substep_overlapped(plan, dt)
                done = 2
                try:
                    self._graph = c.capture(lambda: [self._substep_overlapped(plan, dt) for _ in range(2)])
                    self._graph_dt = dt
                except Exception as e:  # capture not supported by this stack: stay eager
                    self._graph, self.use_graph, self.capture_error = None, False, repr(e)
            if self._graph is not None:
                self.mode = "graph+overlap"
                while done + 2 <= n:
                    c.replay(self._graph)  # two substeps: the field buffers are back in place
                    done += 2
        if self._graph is None:
            self.mode = "overlap"
        for _ in range(n - done):
            self._substep_overlapped(plan, dt)
        self.exchanges += n * nex

    def local_state(self):
        return self.backend.get_state()
