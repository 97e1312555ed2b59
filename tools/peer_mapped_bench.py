"""What the peer-mapped exchange (csrc/comm.hip: pdeopt_comm_ipc_export / _attach) costs per substep: px x py PROCESSES
sharing GPU 0, one rank each (the process-per-GPU deployment folded onto the one GPU a box has), tiles of BASELINE config
5's size.  Every rank times n substeps of pdeopt_rk4_decomposed_advance after a warm-up; a start barrier through the
parent's queues lines the ranks up.  Compare with the same tiles as in-process ranks (bench.py --virtual-ranks) and with
one tile alone (--decomp-grid 2048): the difference is what the wait / publish kernels and the counters cost.
usage: python tools/peer_mapped_bench.py [px py tile_n substeps]"""
import multiprocessing as mp
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def worker(rank, px, py, n, y0, dt, nsub, q_up, q_down, q_out):
    sol = None
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pde_opt_amd as P
        from pde_opt_amd.decomp import CartesianGrid, DecomposedSolver, PeerMappedComm
        from util import MOB, MU, std_domain

        def allgather(obj):
            q_up.put((rank, obj))
            return q_down.get(timeout=300)

        dom = std_domain(P, n * px, n * py)
        eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
        comm = PeerMappedComm(px * py, rank, allgather=allgather)
        sol = DecomposedSolver(eq, CartesianGrid(px, py, rank), comm=comm, dtype=np.float32, halo=8)
        sol.set_global_state(y0)
        sol.advance(dt, 20)  # warm-up
        sol.backend.engine.sync()
        allgather("ready")  # start line
        t0 = time.perf_counter()
        sol.advance(dt, nsub)
        sol.backend.engine.sync()
        el = time.perf_counter() - t0
        tile = sol.local_state()
        q_out.put((rank, el, sol.backend.engine.last_kernel, bool(np.all(np.isfinite(tile)))))
    except BaseException as e:  # noqa: BLE001
        q_out.put((rank, -1.0, repr(e), False))
    finally:
        try:
            while q_down.get(timeout=300) != "done":
                pass
        except BaseException:  # noqa: BLE001
            pass
        if sol is not None:
            sol.backend.engine.close()


def main():
    px, py, n, nsub = (int(v) for v in (sys.argv[1:5] + ["2", "2", "2048", "200"][len(sys.argv) - 1:]))
    world = px * py
    rng = np.random.default_rng(0)
    y0 = np.clip(0.5 + 0.01 * rng.standard_normal((n * px, n * py)), 0.05, 0.95).astype(np.float32)
    ctx = mp.get_context("spawn")
    q_up, q_out = ctx.Queue(), ctx.Queue()
    q_down = [ctx.Queue() for _ in range(world)]
    procs = [ctx.Process(target=worker, args=(r, px, py, n, y0, 2e-7, nsub, q_up, q_down[r], q_out)) for r in range(world)]
    for p_ in procs:
        p_.start()

    def gather():  # two rounds: the hipIpc handles, then the start line
        for _ in range(2):
            got = dict(q_up.get(timeout=300) for _ in range(world))
            for r in range(world):
                q_down[r].put([got[i] for i in range(world)])

    t = threading.Thread(target=gather)
    t.start()
    try:
        res = sorted(q_out.get(timeout=600) for _ in range(world))
    finally:
        t.join(timeout=10)
        for r in range(world):
            q_down[r].put("done")
        for p_ in procs:
            p_.join(timeout=60)
            if p_.is_alive():
                p_.kill()
    worst = max(r[1] for r in res)
    print(f"peer-mapped exchange, {px} x {py} processes on one GPU, {n}^2 tiles, {nsub} substeps: "
          f"{1e6 * worst / nsub:.1f} us per substep (slowest rank; per rank {[round(1e6 * r[1] / nsub, 1) for r in res]}), "
          f"kernel {res[0][2]}, finite {all(r[3] for r in res)}")


if __name__ == "__main__":
    main()
