"""Generate tests/golden/*.npz from the reference's own source.  BUILD-CONTAINER ONLY.

The reference (``/root/reference``, acoh64/pde-opt @ 2025-09-26) is pure Python on
JAX; JAX/diffrax/equinox/gymnasium are not installed here (ordinary
``ModuleNotFoundError`` -- nothing was refused).  Its array code only uses
``jnp.roll/fft/exp/abs/stack/sum/sqrt/meshgrid/linspace/ones_like`` + arithmetic,
whose numpy namesakes have identical semantics, so the reference *source files*
are executed unmodified, loaded by path, with ``jax.numpy`` bound to ``numpy``
(SURVEY.md Appendix B).  Nothing from the reference is copied into this repo:
the files are read where they lie and only arrays (inputs + outputs) are saved.

What this is NOT: a run of real JAX/XLA or of diffrax.  Time stepping loops
(``diffrax.diffeqsolve``) are third-party and absent; goldens for trajectories
call the reference's own ``solver.step`` bodies in a plain Python loop.

Run:  python oracle/gen_golden.py          (no-op when /root/reference is absent)
"""

from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name):
    m = _mod(name)
    m.__path__ = []
    return m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    """Bind stand-ins for the absent third-party modules, then exec the reference files."""
    jax = _mod("jax", numpy=np, jit=lambda f, **k: f, Array=np.ndarray)
    sys.modules["jax.numpy"] = np
    jax.numpy = np
    _mod("equinox", Module=object, filter_jit=lambda f: f)

    class _Results:
        successful = 0

    class _ODETerm:  # dfx.ODETerm(vector_field): keeps the function so that a caller can evaluate it
        def __init__(self, vector_field=None):
            self.vector_field = vector_field

    def _capture_solve(term, solver, t0=None, t1=None, dt0=None, y0=None, **kw):
        """stand-in for dfx.diffeqsolve: no time stepping here (diffrax is absent); records the term and returns
        the initial state, which lets Shape.__post_init__ (shapes.py:34-37) finish and hands its RHS closure out"""
        _capture_solve.last = types.SimpleNamespace(term=term, t0=t0, t1=t1, dt0=dt0, y0=y0, kw=kw)
        return types.SimpleNamespace(ys=[y0])

    _mod(
        "diffrax",
        AbstractSolver=object,
        ODETerm=_ODETerm,
        LocalLinearInterpolation=object,
        RESULTS=_Results,
        diffeqsolve=_capture_solve,
        Tsit5=lambda *a, **k: None,
        PIDController=lambda *a, **k: None,
        SaveAt=lambda *a, **k: None,
    )
    for p in (
        "pde_opt",
        "pde_opt.numerics",
        "pde_opt.numerics.utils",
        "pde_opt.numerics.equations",
        "pde_opt.numerics.symbolic",
    ):
        _pkg(p)
    _mod("pde_opt.numerics.shapes", Shape=object)

    ns = types.SimpleNamespace()
    ns.derivatives = _load(
        "pde_opt.numerics.utils.derivatives", "pde_opt/numerics/utils/derivatives.py"
    )
    ns.domains = _load("pde_opt.numerics.domains", "pde_opt/numerics/domains.py")
    _load("pde_opt.numerics.equations.base_eq", "pde_opt/numerics/equations/base_eq.py")
    ns.ch = _load(
        "pde_opt.numerics.equations.cahn_hilliard", "pde_opt/numerics/equations/cahn_hilliard.py"
    )
    ns.ac = _load(
        "pde_opt.numerics.equations.allen_cahn", "pde_opt/numerics/equations/allen_cahn.py"
    )
    ns.gpe = _load(
        "pde_opt.numerics.equations.gross_pitaevskii",
        "pde_opt/numerics/equations/gross_pitaevskii.py",
    )
    _load("pde_opt.numerics.symbolic.base_sym_eq", "pde_opt/numerics/symbolic/base_sym_eq.py")
    ns.ch_sym = _load(
        "pde_opt.numerics.symbolic.cahn_hilliard_sym",
        "pde_opt/numerics/symbolic/cahn_hilliard_sym.py",
    )
    ns.ac_sym = _load(
        "pde_opt.numerics.symbolic.allen_cahn_sym", "pde_opt/numerics/symbolic/allen_cahn_sym.py"
    )
    ns.solvers = _load("pde_opt.numerics.solvers", "pde_opt/numerics/solvers.py")
    ns.rl_utils = _load("pde_opt.rl_utils", "pde_opt/rl_utils.py")
    # the real shapes.py last (the equations above only needed the name `Shape`); its diffeqsolve call lands in
    # _capture_solve
    ns.shapes = _load("pde_opt.numerics.shapes", "pde_opt/numerics/shapes.py")
    ns.capture_solve = _capture_solve
    return ns


class _AtSetter:
    def __init__(self, arr, idx):
        self._arr, self._idx = arr, idx

    def set(self, v):
        out = self._arr.copy()
        np.ndarray.__setitem__(out, self._idx, v)
        return out


class _At:
    def __init__(self, arr):
        self._arr = arr

    def __getitem__(self, idx):
        return _AtSetter(self._arr, idx)


class JaxLikeArray(np.ndarray):
    """numpy array with jax's functional-update spelling ``a.at[idx].set(v)`` -- the one piece of
    jax.Array API the smoothed-boundary equations use on their inputs (allen_cahn.py:135)."""

    @property
    def at(self):
        return _At(self)


def sbm_geometry(nx, ny, hx, hy, floor=0.05):
    """A smooth level-set field in (floor, 1]: a disc of radius 0.3 min(Lx, Ly), tanh profile."""
    x = (np.arange(nx) + 0.5) * hx
    y = (np.arange(ny) + 0.5) * hy
    X, Y = np.meshgrid(x, y, indexing="ij")
    r = np.sqrt((X - 0.5 * nx * hx) ** 2 + (Y - 0.5 * ny * hy) ** 2)
    eps = 2.5 * max(hx, hy)
    psi = 0.5 * (1.0 + np.tanh((0.3 * min(nx * hx, ny * hy) - r) / eps))
    return floor + (1.0 - floor) * psi


SBM_F = lambda c: c * np.log(c) + (1.0 - c) * np.log(1.0 - c) + 3.0 * c * (1.0 - c) + 0.059  # noqa: E731
SBM_THETA = lambda t: 34.9065850398866 * t**2 - 10.4719755119660 * t + np.pi / 2  # noqa: E731
SBM_FLUX = lambda t: 0.02 * (1.0 + 3.0 * t)  # noqa: E731


class _Terms:
    def __init__(self, vf):
        self._vf = vf

    def vf(self, t, y, args):
        return self._vf(t, y)


def _make(cls, **fields):
    obj = object.__new__(cls)
    for k, v in fields.items():
        setattr(obj, k, v)
    return obj


# closure families used across the goldens
MU = {
    "cubic": lambda c: c**3 - c,
    "regsol": lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c),
}
MOB = {
    "one": lambda c: np.ones_like(c),
    "c1mc": lambda c: c * (1 - c),
    "one_plus_sq": lambda c: 1 + c**2,
    "const015": lambda c: 0.15 * np.ones_like(c),
}


def main():
    if not os.path.isdir(REF):
        print("reference tree absent; nothing to do")
        return
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    Domain = ref.domains.Domain
    rng = np.random.default_rng(20251003)

    # ---- Domain meshes -----------------------------------------------------
    d = Domain((8, 6), ((-1.0, 1.0), (0.0, 3.0)), "dimensionless")
    ax = d.axes()
    fx = d.fft_axes()
    X, Y = d.mesh()
    np.savez(
        os.path.join(OUT, "domain_8x6.npz"),
        ax0=ax[0], ax1=ax[1], f0=fx[0], f1=fx[1], X=X, Y=Y, dx=np.array(d.dx), L=np.array(d.L),
    )

    # ---- RHS goldens ---------------------------------------------------------
    cases = {}
    for (nx, ny) in ((32, 32), (64, 48), (128, 128), (256, 1), (2, 2), (1, 5), (3, 4)):
        for dtype in (np.float64, np.float32):
            lx, ly = 0.01 * nx, 0.01 * ny
            dom = Domain((nx, ny), ((-lx / 2, lx / 2), (-ly / 2, ly / 2)), "dimensionless")
            u_sym = (0.1 * rng.standard_normal((nx, ny))).astype(dtype)
            u_c = np.clip(0.5 + 0.2 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(dtype)
            tag = f"{nx}x{ny}_{np.dtype(dtype).name}"
            kappa = 0.002
            combos = (
                ("cubic", "one", u_sym),
                ("cubic", "one_plus_sq", u_sym),
                ("regsol", "c1mc", u_c),
                ("cubic", "const015", u_sym),
            )
            if nx * ny > 4096:  # keep the fixture file small: one combo at 128^2
                combos = combos[2:3]
            for mu_name, mob_name, u in combos:
                eq = ref.ch.CahnHilliard2DPeriodic(dom, kappa, MU[mu_name], MOB[mob_name], derivs="fd")
                cases[f"ch_fd/{mu_name}/{mob_name}/{tag}/u"] = u
                cases[f"ch_fd/{mu_name}/{mob_name}/{tag}/rhs"] = np.asarray(eq.rhs(u, 0.0))
                eq = ref.ac.AllenCahn2DPeriodic(dom, kappa, MU[mu_name], MOB[mob_name], derivs="fd")
                cases[f"ac_fd/{mu_name}/{mob_name}/{tag}/u"] = u
                cases[f"ac_fd/{mu_name}/{mob_name}/{tag}/rhs"] = np.asarray(eq.rhs(u, 0.0))
            if dtype is np.float64 and 1024 <= nx * ny <= 4096:
                # spectral constants are f64 in numpy but f32 in JAX-without-x64:
                # spectral goldens only in fp64
                eq = ref.ch.CahnHilliard2DPeriodic(dom, kappa, MU["regsol"], MOB["c1mc"], derivs="fourier")
                cases[f"ch_fourier/regsol/c1mc/{tag}/u"] = u_c
                cases[f"ch_fourier/regsol/c1mc/{tag}/rhs"] = np.asarray(eq.rhs(u_c, 0.0))
                cases[f"ch_fourier/regsol/c1mc/{tag}/symbol"] = np.asarray(eq.fourier_symbol)
                eq = ref.ac.AllenCahn2DPeriodic(dom, kappa, MU["cubic"], MOB["one_plus_sq"], derivs="fourier")
                cases[f"ac_fourier/cubic/one_plus_sq/{tag}/u"] = u_sym
                cases[f"ac_fourier/cubic/one_plus_sq/{tag}/rhs"] = np.asarray(eq.rhs(u_sym, 0.0))
    np.savez_compressed(os.path.join(OUT, "rhs_cases.npz"), **cases)

    # ---- manufactured solution (tests/test_rhs_convergence.py) ---------------
    import sympy as sp

    x, y, t = sp.symbols("x y t", real=True)
    u_star = sp.sin(2 * x) * sp.cos(3 * y) * sp.exp(-0.7 * t)
    man = {}
    L = 2 * np.pi
    for n in (32, 64):
        dom = Domain((n, n), ((-L / 2, L / 2), (-L / 2, L / 2)), "dimensionless")
        s = ref.ch_sym.SymbolicCahnHilliard2DPeriodic(
            dom, 1e-2, lambda u: u**3 - u, lambda u: 1 + u**2, u_star
        )
        man[f"ch/{n}/u"] = np.asarray(s.u_exact(0))
        man[f"ch/{n}/rhs_exact"] = np.asarray(s.rhs_exact(0))
        eq = ref.ch.CahnHilliard2DPeriodic(dom, 1e-2, lambda u: u**3 - u, lambda u: 1 + u**2)
        man[f"ch/{n}/rhs_fd"] = np.asarray(eq.rhs(man[f"ch/{n}/u"], 0))
        s = ref.ac_sym.SymbolicAllenCahn2DPeriodic(
            dom, 1e-2, lambda u: u**3 - u, lambda u: 1 + u**2, u_star
        )
        man[f"ac/{n}/rhs_exact"] = np.asarray(s.rhs_exact(0))
        eq = ref.ac.AllenCahn2DPeriodic(dom, 1e-2, lambda u: u**3 - u, lambda u: 1 + u**2)
        man[f"ac/{n}/rhs_fd"] = np.asarray(eq.rhs(man[f"ch/{n}/u"], 0))
    np.savez_compressed(os.path.join(OUT, "manufactured.npz"), **man)

    # ---- IMEX trajectory: reference solver.step looped ------------------------
    traj = {}
    nx = ny = 64
    dom = Domain((nx, ny), ((-0.32, 0.32), (-0.32, 0.32)), "dimensionless")
    eq = ref.ch.CahnHilliard2DPeriodic(dom, 0.002, MU["regsol"], MOB["c1mc"], derivs="fd")
    solver = _make(
        ref.solvers.SemiImplicitFourierSpectral,
        A=0.5, fourier_symbol=eq.fourier_symbol, fft=eq.fft, ifft=eq.ifft,
    )
    y = np.clip(0.5 + 0.01 * rng.standard_normal((nx, ny)), 0.05, 0.95)
    traj["imex/y0"] = y
    terms = _Terms(lambda t, yy: eq.rhs(yy, t))
    dt = 1e-6
    seq = []
    for i in range(10):
        y = solver.step(terms, i * dt, (i + 1) * dt, y, None, None, False)[0]
        seq.append(np.asarray(y))
    traj["imex/ys"] = np.stack(seq)
    traj["imex/dt"] = np.array(dt)

    # 1-D style (256 x 1) IMEX, the tests/test_solvers.py:21-61 set-up, 200 steps
    nx, ny = 256, 1
    dom = Domain((nx, ny), ((-1.28, 1.28), (-0.005, 0.005)), "dimensionless")
    eq = ref.ch.CahnHilliard2DPeriodic(dom, 0.002, MU["cubic"], MOB["one"], derivs="fd")
    solver = _make(
        ref.solvers.SemiImplicitFourierSpectral,
        A=0.5, fourier_symbol=eq.fourier_symbol, fft=eq.fft, ifft=eq.ifft,
    )
    y = np.ones((nx, ny))
    y[: nx // 2, :] = -1.0
    traj["imex1d/y0"] = y
    terms = _Terms(lambda t, yy: eq.rhs(yy, t))
    dt = 5e-5
    for i in range(200):
        y = solver.step(terms, i * dt, (i + 1) * dt, y, None, None, False)[0]
    traj["imex1d/y200"] = np.asarray(y)

    # ---- Strang trajectory (GPE), both the committed A_term == 0 and a real one --
    n = 48
    dom = Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
    geq = ref.gpe.GPE2DTSControl(dom, 1000.0, 0.1, lambda tt, xx, yy: 0.05 * xx, trap_factor=1.0)
    Xm, Ym = dom.mesh()
    psi0 = np.exp(-(Xm**2 + Ym**2) / (2 * 4.0**2)) * np.exp(0.3j * Xm)
    psi0 = psi0 / np.sqrt(np.sum(np.abs(psi0) ** 2) * dom.dx[0] ** 2)
    y0 = np.stack([psi0.real, psi0.imag], axis=-1)
    traj["strang/y0"] = y0
    a_real = 0.5j * geq.two_pi_i_k_2
    for name, a_term, tscale in (
        ("zeroA_imag", geq.A_term, -1j),
        ("realA_real", a_real, 1.0),
        ("realA_imag", a_real, -1j),
    ):
        solver = _make(
            ref.solvers.StrangSplitting,
            A_term=a_term, dx=geq.dx, fft=geq.fft, ifft=geq.ifft, time_scale=tscale,
        )
        terms = _Terms(lambda t, yy: geq.B_terms(yy, t))
        y = y0
        seq = []
        dt = 1e-3
        for i in range(5):
            y = solver.step(terms, i * dt, (i + 1) * dt, y, None, None, False)[0]
            seq.append(np.asarray(y))
        traj[f"strang/{name}/ys"] = np.stack(seq)
    traj["strang/b_terms"] = np.asarray(geq.B_terms(y0, 0.0))
    traj["strang/A_real"] = a_real
    np.savez_compressed(os.path.join(OUT, "trajectories.npz"), **traj)
    # ---- smoothed-boundary equations -----------------------------
    # notebooks/smooth_boundary.ipynb closures; theta(t) is that notebook's quadratic ramp.
    sbm = {}
    for kind, (nx, ny) in (("ac", (64, 128)), ("ch", (96, 64)), ("ac", (40, 100)), ("ch", (50, 24))):
        for dtype in (np.float64, np.float32):
            dom0 = Domain((nx, ny), ((0.0, 1.0 * nx), (0.0, 1.0 * ny)), "dimensionless")
            psi = sbm_geometry(nx, ny, *dom0.dx).astype(dtype).view(JaxLikeArray)
            dom = Domain((nx, ny), dom0.box, "dimensionless", geometry=types.SimpleNamespace(smooth=psi))
            u = np.clip(0.5 + 0.2 * rng.standard_normal((nx, ny)), 0.05, 0.95).astype(dtype)
            kappa = 1.5
            tag = f"{kind}/{nx}x{ny}_{np.dtype(dtype).name}"
            if kind == "ac":
                eq = ref.ac.AllenCahn2DSmoothedBoundary(dom, kappa, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA)
            else:
                eq = ref.ch.CahnHilliard2DSmoothedBoundary(dom, kappa, SBM_F, MU["regsol"], MOB["c1mc"],
                                                           SBM_THETA, SBM_FLUX)
            sbm[f"{tag}/psi"] = np.asarray(psi)
            sbm[f"{tag}/u"] = u
            for t in (0.0, 0.17):
                sbm[f"{tag}/rhs_t{t}"] = np.asarray(eq.rhs(u, t))
            sbm[f"{tag}/norm_grad_psi"] = np.asarray(eq.norm_grad_psi)
            sbm[f"{tag}/left_half"] = np.asarray(eq.left_half)
    np.savez_compressed(os.path.join(OUT, "sbm_cases.npz"), **sbm)

    # ---- rl_utils.detect_vortices ------------------------------------
    vort = {}
    for tag, n, m, dtype in (("48x48_c128", 48, 48, np.complex128), ("40x64_c64", 40, 64, np.complex64)):
        xs = (np.arange(n) + 0.5) - n / 2
        ys = (np.arange(m) + 0.5) - m / 2
        Xv, Yv = np.meshgrid(xs, ys, indexing="ij")
        # a vortex (+1), an antivortex (-1) and a doubly quantised vortex (+2) on a Gaussian cloud + noise
        psi = np.exp(-(Xv**2 + Yv**2) / (2 * (0.3 * n) ** 2)).astype(complex)
        for (cx, cy, q) in ((-6.2, -4.9, 1), (7.3, 3.1, -1), (1.4, -9.6, 2)):
            z = (Xv - cx) + 1j * (Yv - cy)
            psi = psi * (z / np.sqrt(np.abs(z) ** 2 + 1.0)) ** abs(q) if q > 0 else psi * np.conj(z) / np.sqrt(np.abs(z) ** 2 + 1.0)
        psi = psi + 1e-3 * (rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m)))
        psi = psi.astype(dtype)
        vort[f"{tag}/psi"] = psi
        for amp, tol in ((0.0, 0.5), (0.02, 0.5), (0.0, 1.5)):
            r = ref.rl_utils.detect_vortices(psi, amp_thresh=amp, tol=tol)
            key = f"{tag}/amp{amp}_tol{tol}"
            vort[f"{key}/winding"] = np.asarray(r["winding"])
            vort[f"{key}/positions"] = np.asarray(r["positions"])
            vort[f"{key}/charges"] = np.asarray(r["charges"])
            vort[f"{key}/counts"] = np.array([r["num_vortices"], r["total_topological_charge"], r["abs_charge_count"]])
    np.savez_compressed(os.path.join(OUT, "vortices.npz"), **vort)

    # ---- CahnHilliard3DPeriodic ------------------------------------------
    c3 = {}
    for (nx, ny, nz) in ((16, 12, 20), (32, 32, 32), (8, 64, 72)):
        for dtype in (np.float64, np.float32):
            dom3 = Domain((nx, ny, nz), ((-0.005 * nx, 0.005 * nx), (-0.005 * ny, 0.005 * ny), (0.0, 0.012 * nz)), "dimensionless")
            u = np.clip(0.5 + 0.2 * rng.standard_normal((nx, ny, nz)), 0.05, 0.95).astype(dtype)
            eq = ref.ch.CahnHilliard3DPeriodic(dom3, 0.002, MU["regsol"], MOB["c1mc"], derivs="fd")
            tag = f"{nx}x{ny}x{nz}_{np.dtype(dtype).name}"
            c3[f"{tag}/u"] = u
            c3[f"{tag}/rhs"] = np.asarray(eq.rhs(u, 0.0))
            if dtype is np.float64 and nx * ny * nz <= 4096:
                c3[f"{tag}/symbol"] = np.asarray(eq.fourier_symbol)
    np.savez_compressed(os.path.join(OUT, "ch3d_cases.npz"), **c3)

    round2(ref)
    print("wrote goldens to", os.path.abspath(OUT))
    for f in sorted(os.listdir(OUT)):
        print("  ", f, os.path.getsize(os.path.join(OUT, f)))


# lights(t, x, y) of the round-2 goldens: a Gaussian spot that moves and brightens during the solve (the
# shape of an RL stirring control); tests/util.py carries the same expression
MOVING_SPOT = lambda t, x, y: 30.0 * (1.0 + 100.0 * t) * np.exp(-((x + 2.0 - 800.0 * t) ** 2 + (y - 1.0) ** 2) / 4.5)  # noqa: E731


def round2(ref):
    """Fixtures added in round 2.  Own files and an own generator state, so every round-1 file above
    keeps reproducing bit for bit."""
    Domain = ref.domains.Domain
    rng = np.random.default_rng(20251004)

    # ---- Strang with a TIME-DEPENDENT control: the reference evaluates lights(t0, X, Y) in every step
    # (solvers.py:109 -> gross_pitaevskii.py:61,67-75).  48^2 takes the rocFFT pipeline, 64^2 the fused one.
    traj = {}
    for n in (48, 64):
        dom = Domain((n, n), ((-12.0, 12.0), (-12.0, 12.0)), "dimensionless")
        geq = ref.gpe.GPE2DTSControl(dom, 800.0, -0.15, MOVING_SPOT, trap_factor=0.9)
        Xm, Ym = dom.mesh()
        psi0 = np.exp(-(Xm**2 + Ym**2) / (2 * 3.5**2)) * np.exp(-0.2j * Ym)
        psi0 = psi0 / np.sqrt(np.sum(np.abs(psi0) ** 2) * dom.dx[0] ** 2)
        y0 = np.stack([psi0.real, psi0.imag], axis=-1)
        traj[f"strang_tdep/{n}/y0"] = y0
        a_real = 0.5j * geq.two_pi_i_k_2
        for name, tscale in (("real", 1.0), ("imag", -1j)):
            solver = _make(ref.solvers.StrangSplitting, A_term=a_real, dx=geq.dx, fft=geq.fft, ifft=geq.ifft,
                           time_scale=tscale)
            terms = _Terms(lambda t, yy: geq.B_terms(yy, t))
            y, seq, dt = y0, [], 1e-3
            for i in range(6):
                y = solver.step(terms, i * dt, (i + 1) * dt, y, None, None, False)[0]
                seq.append(np.asarray(y))
            traj[f"strang_tdep/{n}/{name}/ys"] = np.stack(seq)
        traj[f"strang_tdep/{n}/b_terms_t0"] = np.asarray(geq.B_terms(y0, 0.0))
        traj[f"strang_tdep/{n}/b_terms_t3"] = np.asarray(geq.B_terms(y0, 3e-3))
    np.savez_compressed(os.path.join(OUT, "trajectories_r2.npz"), **traj)

    # ---- CahnHilliard3DPeriodic.rhs_fourier (cahn_hilliard.py:167-175; 9 transforms), fp64 only (the
    # spectral constants are f64 in numpy but f32 in JAX-without-x64)
    c3 = {}
    for (nx, ny, nz) in ((16, 12, 20), (32, 32, 32), (8, 24, 36)):
        dom3 = Domain((nx, ny, nz), ((-0.005 * nx, 0.005 * nx), (-0.005 * ny, 0.005 * ny), (0.0, 0.012 * nz)), "dimensionless")
        u = np.clip(0.5 + 0.2 * rng.standard_normal((nx, ny, nz)), 0.05, 0.95)
        eq = ref.ch.CahnHilliard3DPeriodic(dom3, 0.002, MU["regsol"], MOB["c1mc"], derivs="fourier")
        tag = f"{nx}x{ny}x{nz}_float64"
        c3[f"{tag}/u"] = u
        c3[f"{tag}/rhs"] = np.asarray(eq.rhs(u, 0.0))
    np.savez_compressed(os.path.join(OUT, "ch3d_fourier.npz"), **c3)

    # ---- Shape (shapes.py:21-203): the smoothing right-hand side of smooth_shape (:44-64, a closure handed to
    # dfx.diffeqsolve -- evaluated here through the capturing stand-in), the graph Laplacian of the mask
    # (:81-143) and the eigenvalues of its lowest modes (:145-203)
    sh = {}
    masks = {}
    X, Y = np.meshgrid(np.arange(48) + 0.5, np.arange(40) + 0.5, indexing="ij")
    masks["disc48x40"] = (((X - 24.0) / 15.0) ** 2 + ((Y - 20.0) / 11.0) ** 2 < 1.0).astype(np.float64)
    m = np.zeros((32, 32))
    m[6:26, 9:21] = 1.0
    m[12:18, 21:29] = 1.0
    masks["tee32"] = m
    for name, mask in masks.items():
        for (dx, eps, curv) in (((1.0, 1.0), 1.0, 0.0), ((0.5, 0.8), 2.0, 0.3), ((1.0, 1.0), 0.7, 1.0)):
            shape = ref.shapes.Shape(mask, dx=dx, smooth_epsilon=eps, smooth_curvature=curv)
            cap = ref.capture_solve.last
            rhs = cap.term.vector_field
            tag = f"{name}/dx{dx[0]}_{dx[1]}_eps{eps}_c{curv}"
            # a smeared field (the binary itself has |grad u| = 0 almost everywhere) and the raw mask
            u = mask.copy()
            for _ in range(3):
                u = 0.2 * (u + np.roll(u, 1, 0) + np.roll(u, -1, 0) + np.roll(u, 1, 1) + np.roll(u, -1, 1))
            u = u + 0.02 * rng.standard_normal(u.shape)
            sh[f"{tag}/u"] = u
            sh[f"{tag}/rhs"] = np.asarray(rhs(0.0, u, None))
            sh[f"{tag}/rhs_binary"] = np.asarray(rhs(0.0, mask, None))
            sh[f"{tag}/solve_args"] = np.array([cap.t0, cap.t1, cap.dt0])
            sh[f"{tag}/post_init_of_y0"] = np.asarray(shape.smooth)  # the clamps of :36-37 applied to ys[-1] = binary
        sh[f"{name}/mask"] = mask
        for periodic in (False, True):
            lap, ids = ref.shapes.Shape(mask).laplacian_from_mask(periodic=periodic)
            sh[f"{name}/laplacian_{'periodic' if periodic else 'open'}"] = lap.toarray()
            sh[f"{name}/ids"] = ids
        shp = ref.shapes.Shape(mask)
        shp.get_shape_modes(6)
        sh[f"{name}/mode_evals"] = np.asarray(shp.shape_basis_evals)
        sh[f"{name}/mode_basis_abs_sum"] = np.abs(np.asarray(shp.shape_basis)).sum(axis=(0, 1))
    np.savez_compressed(os.path.join(OUT, "shapes.npz"), **sh)


if __name__ == "__main__":
    main()
