"""Test-only helpers for the domain-decomposition tests: a CPU tile backend built on the oracle and
an in-process communicator that plays several ranks on one GPU."""
import numpy as np

from oracle import np_oracle as O
from pde_opt_amd.decomp import FROM, HALO, strip_layout


class OracleTileBackend:
    """A rank's padded tile advanced with the numpy oracle (per-stage plan, 4 exchanges/substep).
    The RHS is evaluated with the oracle's periodic stencils on the PADDED array: the wrap-around
    only contaminates cells within 2 of the padded border, the interior (inset 4) is exact."""

    on_device = False
    device = 0

    def __init__(self, tile_shape, hx, hy, kappa, mu, mob, dtype=np.float64):
        self.nx, self.ny = tile_shape
        self.h = HALO
        self.dtype = np.dtype(dtype)
        self.hx, self.hy, self.kappa, self.mu, self.mob = hx, hy, kappa, mu, mob
        shape = (self.nx + 2 * self.h, self.ny + 2 * self.h)
        self.f = [np.zeros(shape, dtype) for _ in range(4)]  # Y, TA, TB, ACC
        self.layout, self.strip_elems = strip_layout(self.nx, self.ny, self.h)

    def _int(self, a):
        h = self.h
        return a[h:-h, h:-h]

    def phase_plan(self):
        return [0, 1, 2, 1]

    def set_state(self, tile):
        self._int(self.f[0])[...] = tile

    def get_state(self):
        return self._int(self.f[0]).copy()

    def _src_pieces(self, field):
        a = self._int(self.f[field])
        h = self.h
        return [a[:h, :], a[-h:, :], a[:, :h], a[:, -h:], a[:h, :h], a[:h, -h:], a[-h:, :h], a[-h:, -h:]]

    def pack(self, field, send):
        buf = send.numpy()
        for (off, shp), piece in zip(self.layout, self._src_pieces(field)):
            buf[off: off + shp[0] * shp[1]] = piece.ravel()

    def unpack(self, field, recv, neighbours):
        buf = recv.numpy().reshape(-1, self.strip_elems)
        a, h, nx, ny = self.f[field], self.h, self.nx, self.ny
        dst = [a[:h, h:h + ny], a[h + nx:, h:h + ny], a[h:h + nx, :h], a[h:h + nx, h + ny:],
               a[:h, :h], a[:h, h + ny:], a[h + nx:, :h], a[h + nx:, h + ny:]]
        for q in range(8):
            off, shp = self.layout[FROM[q]]
            dst[q][...] = buf[neighbours[q], off: off + shp[0] * shp[1]].reshape(shp)

    def _k(self, field):
        return O.ch_rhs_fd(self.f[field], self.hx, self.hy, self.kappa, self.mu, self.mob)

    def phase(self, ph, dt):
        Y, TA, TB, ACC = self.f
        if ph == 0:
            k = self._k(0); TA[...] = Y + (dt / 2) * k; ACC[...] = Y + (dt / 6) * k
        elif ph == 1:
            k = self._k(1); TB[...] = Y + (dt / 2) * k; ACC[...] = ACC + (dt / 3) * k
        elif ph == 2:
            k = self._k(2); TA[...] = Y + dt * k; ACC[...] = ACC + (dt / 3) * k
        else:
            k = self._k(1); Y[...] = ACC + (dt / 6) * k


class _DevBuf:
    def __init__(self, ptr):
        self.ptr = ptr

    def data_ptr(self):
        return self.ptr


class InProcessComm:
    """Several ranks of one process grid living on ONE GPU (one engine each): the all-gather is a
    device-to-device copy of every rank's strip into every rank's receive buffer."""

    def __init__(self, world):
        self.world = world
        self.rank = None
        self.members = []  # (engine, send, recv, nbytes)

    def view(self, rank):
        v = _View(self, rank)
        return v


class _View:
    def __init__(self, parent, rank):
        self.parent, self.rank, self.world = parent, rank, parent.world

    def make_buffers(self, backend):
        nbytes = backend.strip_elems * backend.dtype.itemsize
        eng = backend.engine
        send = _DevBuf(eng.buffer_alloc(nbytes))
        recv = _DevBuf(eng.buffer_alloc(nbytes * self.world))
        self.parent.members.append((self.rank, eng, send, recv, nbytes))
        return send, recv

    def all_gather(self, send, recv):
        pass  # performed collectively by InProcessComm.gather_all


def gather_all(comm):
    from pde_opt_amd import _lib as L

    for _, eng, _, _, _ in comm.members:
        eng.sync()
    for _, eng_dst, _, recv, nbytes in comm.members:
        for src_rank, _, send_src, _, _ in comm.members:
            eng_dst.buffer_copy(recv.ptr + src_rank * nbytes, send_src.ptr, nbytes, L.COPY_D2D)


# ---------------------------------------------------------------------------------------------------------------------
# peer-mapped exchange between PROCESSES (csrc/comm.hip: pdeopt_comm_ipc_export / _attach): worker of
# tests/test_gpu_decomp.py::test_peer_mapped_exchange_between_processes_sharing_one_gpu
def peer_mapped_worker(rank, px, py, shape, y0, dt, calls, dtype_name, q_up, q_down, q_out, sabotage=False):
    """one rank = one process with its own engine on GPU 0; the 64-byte hipIpc handles travel through the parent.  A rank
    keeps its engine (= the buffers its peers have mapped and may still be polling) alive until the parent says every rank
    has reported -- also when it failed or plays the absent rank."""
    sol = None
    try:
        import os
        import sys

        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, os.path.dirname(here))
        sys.path.insert(0, here)
        import numpy as np

        import pde_opt_amd as P
        from pde_opt_amd.decomp import CartesianGrid, DecomposedSolver, PeerMappedComm
        from util import MOB, MU, std_domain

        def allgather(obj):
            q_up.put((rank, obj))
            return q_down.get(timeout=120)

        nx, ny = shape
        dom = std_domain(P, nx, ny)
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        dtype = np.dtype(dtype_name).type
        comm = PeerMappedComm(px * py, rank, allgather=allgather)
        sol = DecomposedSolver(eq, CartesianGrid(px, py, rank), comm=comm, dtype=dtype, halo=8)
        sol.set_global_state(y0)
        if sabotage and rank == px * py - 1:
            q_out.put((rank, "absent", None, None))  # this rank never joins the substep loop
        else:
            for n in calls:
                sol.advance(dt, n)
            q_out.put((rank, sol.mode, sol.backend.engine.last_kernel, sol.local_state()))
    except BaseException as e:  # noqa: BLE001
        q_out.put((rank, "error", repr(e), None))
    finally:
        try:
            while q_down.get(timeout=120) != "done":  # (the handle list, if the failure came before it was consumed)
                pass
        except BaseException:  # noqa: BLE001
            pass
        if sol is not None:
            sol.backend.engine.close()
