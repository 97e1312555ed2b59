"""Domain decomposition of ONE large periodic field over a px x py grid of ranks (BASELINE config 5:
Cahn-Hilliard 4096^2 on 2x2 GPUs).  New relative to the reference, which has no multi-device code
(SURVEY 2.1); semantics = the monolithic periodic solve, bit for bit for the explicit integrators.

Every rank owns an (nx/px) x (ny/py) tile stored with a 4-cell halo.  Per RK4 substep::

    for phase, field in enumerate(backend.phase_plan()):       # 2 phases (fused pairs) or 4
        backend.pack(field, send)                              # 8 interior pieces -> one strip
        comm.all_gather(send, recv)                            # RCCL over xGMI (torch.distributed)
        backend.unpack(field, recv, grid.neighbours())         # 8 halo pieces from 8 neighbours
        backend.phase(phase, dt)                               # fused stencil + RK update kernel

The exchange is ONE all-gather of packed strips per phase (edges and corners together), the pattern
BASELINE.json names; strips are 2*h*(nx+ny)+4*h^2 elements (128 KiB at 2048^2 fp32), far below the
per-link xGMI bandwidth, so the collective is latency-bound: fewer, fatter exchanges (halo 4 for a
fused stage pair) is the lever, not bandwidth.  IMEX / Strang would need a distributed FFT
(all-to-all transposes): replicas only for the spectral integrators.

``TileBackend`` is the HIP engine by default; tests inject a CPU backend built on the oracle to
check the exchange protocol under gloo (world_size 2 and 4) without a GPU.
"""

from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L
from .engine import HipEngine

HALO = 4


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
        if dist.get_backend(group) == "nccl":
            self.stream = torch.cuda.Stream()
            self.stream_handle = self.stream.cuda_stream

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


# ---------------------------------------------------------------------------------- HIP backend
class HipTileBackend:
    """One rank's tile on the GPU (padded layout of libpdeopt_hip.so)."""

    on_device = True

    def __init__(self, equation, tile_shape, dtype=np.float32, device: int = 0, stream: Optional[int] = None,
                 engine: Optional[HipEngine] = None):
        self.engine = engine or HipEngine(device, stream=stream)
        self.device = device
        self.dtype = np.dtype(dtype)
        prob = dict(equation._engine_problem())
        prob["nx"], prob["ny"] = tile_shape  # same spacing, local extent
        self.engine.set_halo_layout(HALO)
        self.engine.configure(dtype=dtype, batch=1, **prob)
        self.engine.set_halo_layout(0)  # the option only applies to the configure above
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

    def phase(self, phase: int, dt: float):
        self.engine.rk4_phase(phase, dt)

    def set_state(self, tile):
        self.engine.set_state(np.asarray(tile, dtype=self.dtype))

    def get_state(self):
        return self.engine.get_state()[0]


# ---------------------------------------------------------------------------------- driver
class DecomposedSolver:
    """RK4 on one rank's tile of a decomposed periodic field."""

    def __init__(self, equation, grid: CartesianGrid, comm=None, dtype=np.float32, device: int = 0,
                 backend=None, stream: Optional[int] = None):
        self.equation = equation
        self.grid = grid
        self.comm = comm or LoopbackComm()
        if self.comm.world != grid.world:
            raise ValueError(f"communicator has {self.comm.world} ranks, process grid {grid.world}")
        nx, ny = equation.domain.points
        self.tile_shape = grid.tile_shape(nx, ny)
        if min(self.tile_shape) < 2 * HALO:
            raise ValueError("tiles must be at least 8 cells wide")
        if stream is None:
            stream = getattr(self.comm, "stream_handle", None)
        self.backend = backend or HipTileBackend(equation, self.tile_shape, dtype, device, stream)
        self.send, self.recv = self.comm.make_buffers(self.backend)
        self.neighbours = grid.neighbours()
        self.exchanges = 0

    def set_global_state(self, u_global):
        si, sj = self.grid.tile_slices(*self.equation.domain.points)
        self.backend.set_state(np.asarray(u_global)[si, sj])

    def exchange(self, field: int):
        self.backend.pack(field, self.send)
        self.comm.all_gather(self.send, self.recv)
        self.backend.unpack(field, self.recv, self.neighbours)
        self.exchanges += 1

    def advance(self, dt: float, n_substeps: int):
        plan = self.backend.phase_plan()
        for _ in range(int(n_substeps)):
            for phase, field in enumerate(plan):
                self.exchange(field)
                self.backend.phase(phase, dt)

    def local_state(self):
        return self.backend.get_state()
