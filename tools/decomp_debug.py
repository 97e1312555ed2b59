"""debug: which ingredient of the graph-replayed decomposed substep goes wrong on one rank"""
import os, socket, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, torch.distributed as dist
import pde_opt_amd as P
from pde_opt_amd import _lib as L
from pde_opt_amd.decomp import CartesianGrid, DecomposedSolver, TorchComm
from util import MOB, MU, std_domain

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
nx, ny = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(4)
dom = std_domain(P, nx, ny)
eq = P.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"])
y0 = np.clip(0.5 + 0.05 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(np.float32)
N = 12
eng = P.HipEngine(); want = P.diffeqsolve(eq, P.RK4(), 0.0, N * 2e-7, 2e-7, y0, engine=eng).ys[-1]; eng.close()
comm = TorchComm()

def report(tag, got):
    d = np.abs(got - want)
    rows = np.where(d.max(axis=1) > 0)[0]; cols = np.where(d.max(axis=0) > 0)[0]
    print(f"{tag:40s} max diff {d.max():.3e} rows {rows[:6]}..{rows[-3:] if len(rows) else ''} n={len(rows)} cols n={len(cols)} {cols[:4]}..{cols[-3:] if len(cols) else ''}", flush=True)

def body_variant(sol, dt, variant):
    be, c = sol.backend, sol.comm
    plan = be.phase_plan()
    for phase, field in enumerate(plan):
        be.pack(field, sol.send)
        if variant == "same_stream":       # collective on the engine's stream, no events
            c.all_gather(sol.send, sol.recv)
            be.unpack(field, sol.recv, sol.neighbours); be.phase(phase, dt)
        elif variant == "copy_same_stream":  # no RCCL: torch copy on the engine's stream
            with torch.cuda.stream(c.stream):
                sol.recv.copy_(sol.send)
            be.unpack(field, sol.recv, sol.neighbours); be.phase(phase, dt)
        elif variant == "copy_other_stream":  # no RCCL: torch copy on the comm stream with the events
            c.ev_packed.record(c.stream); c.comm_stream.wait_event(c.ev_packed)
            with torch.cuda.stream(c.comm_stream):
                sol.recv.copy_(sol.send)
            c.ev_gathered.record(c.comm_stream)
            be.phase(phase, dt, L.PART_INTERIOR); c.wait_gathered()
            be.unpack(field, sol.recv, sol.neighbours); be.phase(phase, dt, L.PART_EDGE)
        elif variant == "rccl_other_stream":
            c.all_gather_overlapped(sol.send, sol.recv)
            be.phase(phase, dt, L.PART_INTERIOR); c.wait_gathered()
            be.unpack(field, sol.recv, sol.neighbours); be.phase(phase, dt, L.PART_EDGE)
        elif variant == "rccl_other_stream_nosplit":
            c.all_gather_overlapped(sol.send, sol.recv); c.wait_gathered()
            be.unpack(field, sol.recv, sol.neighbours); be.phase(phase, dt)

for variant in ("same_stream", "copy_same_stream", "copy_other_stream", "rccl_other_stream_nosplit", "rccl_other_stream"):
    for use_graph in (False, True):
        sol = DecomposedSolver(eq, CartesianGrid(1, 1, 0), comm=comm, dtype=np.float32)
        sol.set_global_state(y0)
        dt = 2e-7
        try:
            if use_graph:
                for _ in range(2): body_variant(sol, dt, variant)
                g = comm.capture(lambda: [body_variant(sol, dt, variant) for _ in range(2)])
                for _ in range((N - 2) // 2): comm.replay(g)
            else:
                for _ in range(N): body_variant(sol, dt, variant)
            torch.cuda.synchronize()
            report(f"{variant} graph={use_graph}", sol.local_state())
        except Exception as e:
            print(f"{variant} graph={use_graph} FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
for n in (4, 5, 12, 13, 50, 51):
    eng = P.HipEngine(); w = P.diffeqsolve(eq, P.RK4(), 0.0, n * 2e-7, 2e-7, y0, engine=eng).ys[-1]; eng.close()
    sol = DecomposedSolver(eq, CartesianGrid(1, 1, 0), comm=comm, dtype=np.float32)
    sol.set_global_state(y0)
    sol.advance(2e-7, n)
    torch.cuda.synchronize()
    got = sol.local_state()
    print("advance n =", n, sol.mode, "max diff", float(np.abs(got - w).max()), flush=True)
    # which substep count does it equal?
    for m in range(max(1, n - 4), n + 5):
        eng = P.HipEngine(); wm = P.diffeqsolve(eq, P.RK4(), 0.0, m * 2e-7, 2e-7, y0, engine=eng).ys[-1]; eng.close()
        if np.array_equal(wm, got):
            print("   == monolithic after", m, "substeps")
dist.destroy_process_group()
