"""``diffeqsolve``: the stepping loop of the reference (third-party ``diffrax.diffeqsolve``, call
sites pde_opt/pde_env.py:293-303, pde_opt/pde_model.py:120-134) driven on the GPU.

Constant stepping: ``n_full`` substeps of exactly ``dt0`` plus one clipped final substep, issued
as at most a few ``pdeopt_advance`` calls (the per-substep loop lives in the library, so an
environment step costs one Python->C transition, not one per substep).  ``SaveAt(ts=...)`` uses
the linear dense output the reference's custom solvers declare (``LocalLinearInterpolation``,
numerics/solvers.py:48,91): steps never stop at save points.

Adaptive stepping (``Tsit5`` + ``PIDController``): trial steps and the scaled RMS error norm run
on the GPU (``pdeopt_tsit5_trial``); the scalar PID update runs here.  With a batch, all
environments share the step (the controller sees the worst error norm).  ``Tsit5`` save points inside
a step use its 4th-order dense output (``pdeopt_tsit5_dense``), as diffrax.Tsit5 does.
"""

from __future__ import annotations

import dataclasses
import math
from typing import Optional

import numpy as np

from . import _lib as L
from .engine import HipEngine
from .numerics.solvers import ConstantStepSize, PIDController, SaveAt


@dataclasses.dataclass
class Solution:
    ts: np.ndarray
    ys: np.ndarray
    stats: dict


def constant_step_plan(t0, t1, dt, rel_tol=1e-9):
    """(n_full, remainder) of ``while t < t1: t = min(t + dt, t1)``; remainders below
    ``rel_tol*dt`` (floating-point dust of the accumulated time upstream) are dropped."""
    span = float(t1) - float(t0)
    if span <= 0:
        return 0, 0.0
    if not dt > 0:
        raise ValueError("dt0 must be positive")
    n_full = int(math.floor(span / dt + rel_tol))
    rem = span - n_full * dt
    if rem <= rel_tol * dt:
        rem = 0.0
    return n_full, rem


def _prepare(equation, solver, y0, engine: Optional[HipEngine], t_aux: float, t_end: Optional[float] = None):
    y0 = np.asarray(y0)
    if np.iscomplexobj(y0):
        raise ValueError("complex states are stored as (..., 2) real/imag pairs (gross_pitaevskii.py:75)")
    if y0.dtype not in (np.float32, np.float64):
        y0 = y0.astype(np.float64)
    npts = len(equation.domain.points)  # 2, or 3 for the 3-D equations
    nd = npts + len(equation._state_trailing)
    single = y0.ndim == nd
    yb = y0[None] if single else y0
    if yb.ndim != nd + 1 or tuple(yb.shape[1:1 + npts]) != tuple(equation.domain.points):
        raise ValueError(f"y0 shape {y0.shape} does not match domain points {equation.domain.points}")
    if engine is None:
        from .engine import default_engine

        engine = default_engine()
    engine.configure(dtype=yb.dtype, batch=yb.shape[0], **equation._engine_problem())
    equation._engine_upload(engine, t_aux, t_end)  # time-dependent terms register a per-substep source
    solver.configure_engine(engine, equation)
    engine.set_state(yb)
    return engine, single


def diffeqsolve(
    equation,
    solver,
    t0,
    t1,
    dt0,
    y0,
    saveat: Optional[SaveAt] = None,
    stepsize_controller=None,
    max_steps: Optional[int] = 1_000_000,
    throw: bool = True,
    engine: Optional[HipEngine] = None,
    args=None,
):
    """Integrate ``equation`` from ``t0`` to ``t1`` on the GPU.  ``y0`` is one field ``(nx, ny)``
    or a batch ``(B, nx, ny)``; ``ys`` gains a leading save axis like ``solution.ys`` upstream."""
    saveat = saveat or SaveAt(t1=True)
    controller = stepsize_controller or ConstantStepSize()
    t0, t1 = float(t0), float(t1)
    eng, single = _prepare(equation, solver, y0, engine, t0, t1)
    take = (lambda a: a[0]) if single else (lambda a: a)

    if isinstance(controller, PIDController):
        return _solve_adaptive(eng, equation, solver, t0, t1, float(dt0), saveat, controller, max_steps, throw, take)

    dt = float(dt0)
    n_full, rem = constant_step_plan(t0, t1, dt)
    total_steps = n_full + (1 if rem > 0 else 0)
    if max_steps is not None and total_steps > max_steps:
        if throw:
            raise RuntimeError(f"max_steps={max_steps} reached ({total_steps} steps needed)")
        total_steps = max_steps
        n_full, rem = min(n_full, max_steps), 0.0

    def advance_steps(first, count):
        """advance `count` steps starting at step index `first`"""
        full = max(0, min(first + count, n_full) - first)
        if full:
            eng.advance(solver.integrator, dt, full, t0 + first * dt)
        if first + count > n_full and rem > 0:
            eng.advance(solver.integrator, rem, 1, t0 + n_full * dt)

    def edge(i):  # time at the end of step i-1 / start of step i
        return t1 if i >= total_steps else t0 + i * dt

    # Dense output inside a step: diffrax.Tsit5 evaluates its own 4th-order interpolant; Euler, RK4 and the
    # reference's custom solvers (LocalLinearInterpolation, solvers.py:48,91) interpolate linearly between
    # the step's end points.
    tsit5 = solver.integrator == L.INT_TSIT5
    ts_out, ys_out = [], []
    done = 0          # steps completed (a pending Tsit5 trial counts once it is committed)
    inside = None     # index (1-based end) of the step whose interior is being sampled

    def finish_inside():
        nonlocal done, inside
        if inside is not None and tsit5:
            eng.tsit5_commit(True)
            done = inside
        inside = None

    def advance_to(k):
        nonlocal done
        finish_inside()
        advance_steps(done, k - done)
        done = k

    if saveat.t0:
        ts_out.append(t0)
        ys_out.append(take(eng.get_state()))
    if saveat.ts is not None:
        for tq in [float(t) for t in saveat.ts]:
            # step index k with edge(k) < tq <= edge(k+1)
            if tq <= t0 or total_steps == 0:
                k_end = 0
            else:
                k_end = min(total_steps, max(1, int(math.ceil((tq - t0) / dt - 1e-9))))
            if k_end == 0 or abs(edge(k_end) - tq) <= 1e-12 * max(1.0, abs(tq)):
                advance_to(k_end)
                ys_out.append(take(eng.get_state()))
            else:
                a, b = edge(k_end - 1), edge(k_end)
                if inside != k_end:  # first save point inside this step
                    advance_to(k_end - 1)
                    if tsit5:
                        eng.tsit5_trial(a, b - a, 1.0, 1.0)  # stages only; committed by finish_inside()
                    else:
                        eng.snapshot()
                        advance_steps(done, 1)
                        done = k_end
                    inside = k_end
                th = (tq - a) / (b - a)
                ys_out.append(take(eng.tsit5_dense(th, b - a) if tsit5 else eng.get_interpolated(th)))
            ts_out.append(tq)
    if saveat.t1 or saveat.ts is None:
        advance_to(total_steps)
        ts_out.append(t1)
        ys_out.append(take(eng.get_state()))
    finish_inside()
    stats = {"num_steps": total_steps, "num_accepted_steps": total_steps, "num_rejected_steps": 0,
             "kernel": eng.last_kernel}
    return Solution(np.asarray(ts_out), np.stack(ys_out), stats)


def _pid_update(c: PIDController, err: float, prev_inv: float, prev_prev_inv: float, order: float = 5.0):
    """diffrax.PIDController's step-size factor for one scaled error norm: ``(keep, factor, 1/err)``"""
    keep = bool(err < 1.0)  # NaN error norm rejects
    inv = 1.0 / err if err > 0 and math.isfinite(err) else (np.inf if err == 0 else 0.0)
    k1 = (c.icoeff + c.pcoeff + c.dcoeff) / order
    k2 = -(c.pcoeff + 2 * c.dcoeff) / order
    k3 = c.dcoeff / order
    f = 1.0
    for base, expo in ((inv, k1), (prev_inv, k2), (prev_prev_inv, k3)):
        if expo != 0.0:
            f *= (base**expo) if math.isfinite(base) and base > 0 else (c.factormax if base > 0 else c.factormin)
    f = min(c.factormax, max(c.factormin, c.safety * f))
    if not keep:
        f = min(1.0, f)
    return keep, f, inv


def _clip_dt(c: PIDController, dt: float) -> float:
    if c.dtmin is not None:
        dt = max(dt, c.dtmin)
    if c.dtmax is not None:
        dt = min(dt, c.dtmax)
    return dt


def _solve_adaptive(eng, equation, solver, t0, t1, dt0, saveat, c: PIDController, max_steps, throw, take):
    if solver.integrator != L.INT_TSIT5:
        raise ValueError("PIDController needs an embedded pair: use Tsit5")
    if (eng.batch == 1 or c.per_environment) and t1 > t0 and dt0 > 0 and getattr(eng, "tsit5_solve_small_supported", lambda: False)():
        return _solve_adaptive_in_kernel(eng, t0, t1, dt0, saveat, c, max_steps, throw, take)
    if c.per_environment and eng.batch > 1:
        return _solve_adaptive_per_env(eng, t0, t1, dt0, saveat, c, max_steps, throw, take, equation)
    t, dt = t0, dt0
    accepted = rejected = 0
    ts_req = [float(v) for v in saveat.ts] if saveat.ts is not None else []
    ts_out, ys_out = [], []
    qi = 0
    if saveat.t0:  # diffrax order: t0, then ts, then t1 (the constant-step driver and the per-environment one agree)
        ts_out.append(t0); ys_out.append(take(eng.get_state()))
    while qi < len(ts_req) and ts_req[qi] <= t0:
        ts_out.append(ts_req[qi]); ys_out.append(take(eng.get_state())); qi += 1
    prev_inv = prev_prev_inv = 1.0
    while t < t1:
        if max_steps is not None and accepted + rejected >= max_steps:
            if throw:
                raise RuntimeError(f"max_steps={max_steps} reached at t={t}")
            break
        h = min(dt, t1 - t)
        err = float(np.max(eng.tsit5_trial(t, h, c.rtol, c.atol)))
        keep, f, inv = _pid_update(c, err, prev_inv, prev_prev_inv)
        if keep:
            accepted += 1
            t_new = t + h
            # dense output from the accepted step's seven slopes, before the commit recycles them:
            # diffrax.Tsit5's 4th-order interpolant (Tsitouras 2011, section 4)
            while qi < len(ts_req) and ts_req[qi] <= t_new + 1e-14 * max(1.0, abs(t_new)):
                th = (ts_req[qi] - t) / h
                ys_out.append(take(eng.tsit5_dense(min(1.0, max(0.0, th)), h)))
                ts_out.append(ts_req[qi]); qi += 1
        eng.tsit5_commit(keep)
        if keep:
            t = t_new if t_new < t1 - 1e-14 * max(1.0, abs(t1)) else t1
            prev_prev_inv, prev_inv = prev_inv, inv
        else:
            rejected += 1
        dt = _clip_dt(c, h * f)
    if saveat.t1 or saveat.ts is None:
        ts_out.append(t); ys_out.append(take(eng.get_state()))
    stats = {"num_steps": accepted + rejected, "num_accepted_steps": accepted, "num_rejected_steps": rejected,
             "kernel": eng.last_kernel}
    return Solution(np.asarray(ts_out), np.stack(ys_out), stats)


_NO_STEP_LIMIT = 100_000_000  # max_steps=None on the in-kernel path: the launch still has to end


def _solve_adaptive_in_kernel(eng, t0, t1, dt0, saveat, c: PIDController, max_steps, throw, take):
    """The adaptive solve of a grid that fits one compute unit's LDS: trial steps, error norms, the PID controller and
    the dense output all run inside ONE launch (``pdeopt_tsit5_solve_small``; csrc/stencil_small_adaptive.hpp), every
    environment with its own controller -- the semantics of ``_solve_adaptive`` for one environment and of
    ``_solve_adaptive_per_env`` for several, without a host round trip per step."""
    B = eng.batch
    ts_req = [float(v) for v in saveat.ts] if saveat.ts is not None else []
    n_t0 = sum(1 for v in ts_req if v <= t0)
    y_start = eng.get_state() if (saveat.t0 or n_t0) else None
    saves, stats = eng.tsit5_solve_small(t0, t1, dt0, c, _NO_STEP_LIMIT if max_steps is None else int(max_steps), ts_req[n_t0:])
    if any(s["status"] == L.TSIT5_STALLED for s in stats):
        bad = next(s for s in stats if s["status"] == L.TSIT5_STALLED)
        raise RuntimeError(f"step size underflow at t={bad['t']} (dt={bad['dt']})")
    if throw and any(s["status"] == L.TSIT5_MAX_STEPS for s in stats):
        bad = next(s for s in stats if s["status"] == L.TSIT5_MAX_STEPS)
        raise RuntimeError(f"max_steps={max_steps} reached at t={bad['t']}")
    ts_out, ys = [], []
    if saveat.t0:
        ts_out.append(t0); ys.append(y_start)
    for q, tq in enumerate(ts_req):
        ts_out.append(tq); ys.append(y_start if q < n_t0 else saves[q - n_t0])
    if B == 1:  # the single-controller driver stops recording where the solve stopped (max_steps with throw=False)
        keep = len(ts_out) - (len(ts_req) - n_t0 - stats[0]["saved"])
        ts_out, ys = ts_out[:keep], ys[:keep]
    if saveat.t1 or saveat.ts is None:
        ts_out.append(max(s["t"] for s in stats)); ys.append(eng.get_state())
    acc, rej = [s["accepted"] for s in stats], [s["rejected"] for s in stats]
    if B == 1:
        st = {"num_steps": acc[0] + rej[0], "num_accepted_steps": acc[0], "num_rejected_steps": rej[0]}
    else:
        st = {"num_steps": max(a + r for a, r in zip(acc, rej)), "num_accepted_steps": acc, "num_rejected_steps": rej}
    st["kernel"] = eng.last_kernel
    return Solution(np.asarray(ts_out), np.stack([take(y) for y in ys]), st)


def _solve_adaptive_per_env(eng, t0, t1, dt0, saveat, c: PIDController, max_steps, throw, take, equation=None):
    """``PIDController(per_environment=True)``: every environment of the batch runs its own controller -- own
    time, own step size, own accept / reject -- exactly as if it were solved alone, while the stages of all
    environments still execute as one batched launch (``pdeopt_tsit5_trial_env``: slopes scaled by
    dt_b / dt_ref).  Environments that have reached ``t1`` idle with dt = 0."""
    B = eng.batch
    if equation is not None and equation._time_dependent_rhs(t0, t1):
        # environments sit at different times, the stage launch carries one: refuse before the first trial step
        # (the library refuses too -- pdeopt_tsit5_trial_env -- but only from inside the loop)
        raise ValueError("PIDController(per_environment=True) needs an autonomous right-hand side: "
                         f"{type(equation).__name__} has time-dependent terms")
    t = np.full(B, t0)
    dt = np.full(B, dt0)
    prev_inv, prev_prev_inv = np.ones(B), np.ones(B)
    accepted, rejected = np.zeros(B, dtype=int), np.zeros(B, dtype=int)
    ts_req = [float(v) for v in saveat.ts] if saveat.ts is not None else []
    n_t0 = sum(1 for v in ts_req if v <= t0)
    y_start = eng.get_state()
    slots = [[y_start[b]] * n_t0 + [None] * (len(ts_req) - n_t0) for b in range(B)]  # per environment, per save point
    qi = np.full(B, n_t0)
    eps1 = 1e-14 * max(1.0, abs(t1))
    while np.any(t < t1):
        active = t < t1
        if max_steps is not None and np.any((accepted + rejected)[active] >= max_steps):
            if throw:
                raise RuntimeError(f"max_steps={max_steps} reached")
            break
        h = np.where(active, np.minimum(dt, t1 - t), 0.0)
        err, h_ref = eng.tsit5_trial_env(float(t[active].min()), h, c.rtol, c.atol)
        keep = np.zeros(B, dtype=bool)
        for b in np.nonzero(active)[0]:
            keep[b], f, inv = _pid_update(c, float(err[b]), prev_inv[b], prev_prev_inv[b])
            if keep[b]:
                accepted[b] += 1
                t_new = t[b] + h[b]
                while qi[b] < len(ts_req) and ts_req[qi[b]] <= t_new + 1e-14 * max(1.0, abs(t_new)):
                    th = min(1.0, max(0.0, (ts_req[qi[b]] - t[b]) / h[b]))
                    slots[b][qi[b]] = eng.tsit5_dense(th, h_ref, env_first=int(b), env_count=1)[0]
                    qi[b] += 1
                t[b] = t_new if t_new < t1 - eps1 else t1
                prev_prev_inv[b], prev_inv[b] = prev_inv[b], inv
            else:
                rejected[b] += 1
            dt[b] = _clip_dt(c, h[b] * f)
        eng.tsit5_commit_env(keep)
    y_end = eng.get_state()
    for b in range(B):  # save points never reached (max_steps with throw=False) are NaN-filled, never a stale state
        for q in range(len(ts_req)):
            if slots[b][q] is None:
                slots[b][q] = np.full_like(y_end[b], np.nan)
    ts_out = list(ts_req)
    ys = [np.stack([slots[b][q] for b in range(B)]) for q in range(len(ts_req))]
    if saveat.t0:  # t0 first, as in the shared-step and constant-step drivers
        ts_out.insert(0, t0); ys.insert(0, y_start)
    if saveat.t1 or saveat.ts is None:
        ts_out.append(float(t.max())); ys.append(y_end)
    stats = {"num_steps": int((accepted + rejected).max()), "num_accepted_steps": accepted.tolist(),
             "num_rejected_steps": rejected.tolist(), "kernel": eng.last_kernel}
    return Solution(np.asarray(ts_out), np.stack(ys), stats)
