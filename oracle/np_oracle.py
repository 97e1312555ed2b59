"""CPU oracle (numpy restatement) of the pde_opt hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``pde_opt_amd``) never imports anything under ``oracle/`` and
fails loudly when the HIP library is missing.

Every function restates, in index form, arithmetic that lives in the reference
tree (``/root/reference`` = acoh64/pde-opt @ 2025-09-26); the cited ``file:line``
is the code it follows.  Arrays are C-order ``(Nx, Ny)``; axis 0 is x (stride
Ny), axis 1 is y (contiguous); all neighbour accesses are periodic.  dtype is
preserved (float32 in -> float32 out) so fp32 trajectories are comparable.

Parity status (see DESIGN.md "Oracle"):
  * RHS operators, IMEX step, Strang step: PINNED against outputs of the
    reference's own source files executed in the build container under a numpy
    stand-in for ``jax.numpy`` (``oracle/gen_golden.py`` -> ``tests/golden``),
    and against the reference's known-answer tests (sympy manufactured
    solution slope, tanh profile, Thomas-Fermi density).
  * Euler / RK4 / Tsit5 / the stepping loop: the algorithm lives in the
    third-party dependency ``diffrax`` (``>=0.7.0``, unpinned, not vendored,
    ``pyproject.toml:17``); restated from the published tableaux.  Pinned only
    through the reference's integrator-independent analytic end states.
  * Advection-diffusion: absent from the reference package -> PARITY UNPINNED
    (SURVEY.md section 8 a15); the discretisation is defined here from the
    reference's own face primitives.
"""

from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------
# periodic neighbour access
# --------------------------------------------------------------------------


def nb(a, d, axis):
    """``out[i] = a[i + d]`` along ``axis``, periodic.

    The reference spells this ``jnp.roll(a, -d, axis)``
    (pde_opt/numerics/utils/derivatives.py:10-11).
    """
    return np.roll(a, -d, axis)


# --------------------------------------------------------------------------
# finite-difference primitives  (pde_opt/numerics/utils/derivatives.py)
# --------------------------------------------------------------------------


def lap5(u, hx, hy):
    """5-point Laplacian, derivatives.py:8-12."""
    ddx = (nb(u, 1, 0) - 2 * u + nb(u, -1, 0)) / hx**2
    ddy = (nb(u, 1, 1) - 2 * u + nb(u, -1, 1)) / hy**2
    return ddx + ddy


def grad_face(a, h, axis):
    """centre -> face (+1/2 along ``axis``) difference, derivatives.py:24-31."""
    return (nb(a, 1, axis) - a) / h


def avg_face(a, axis):
    """centre -> face (+1/2) linear interpolation, derivatives.py:39-46."""
    return 0.5 * (a + nb(a, 1, axis))


def div_face(F, h, axis):
    """face -> centre divergence, derivatives.py:54-61."""
    return (F - nb(F, -1, axis)) / h


def grad_c(a, h, axis):
    """centred first difference, derivatives.py:62-74."""
    return 0.5 * (nb(a, 1, axis) - nb(a, -1, axis)) / h


def grad2_c(a, h, axis):
    """centred second difference, derivatives.py:77-89."""
    return (nb(a, 1, axis) - 2 * a + nb(a, -1, axis)) / h**2


def grad2xy_c(a, hx, hy):
    """centred mixed second difference, derivatives.py:92-99."""
    return (nb(nb(a, 1, 0), 1, 1) + nb(nb(a, -1, 0), -1, 1) - nb(nb(a, 1, 0), -1, 1) - nb(nb(a, -1, 0), 1, 1)) / (4.0 * hx * hy)


# --------------------------------------------------------------------------
# right-hand sides
# --------------------------------------------------------------------------


def chem_potential(u, hx, hy, kappa, mu_h):
    """mu = mu_h(u) - kappa lap(u), cahn_hilliard.py:93 / allen_cahn.py:83."""
    return mu_h(u) - kappa * lap5(u, hx, hy)


def ch_rhs_fd(u, hx, hy, kappa, mu_h, D):
    """Cahn-Hilliard ``div(D(u) grad mu)``, cahn_hilliard.py:89-109."""
    mu = chem_potential(u, hx, hy, kappa, mu_h)
    gx = grad_face(mu, hx, 0)
    gy = grad_face(mu, hy, 1)
    Du = D(u)
    Fx = avg_face(Du, 0) * gx
    Fy = avg_face(Du, 1) * gy
    return div_face(Fx, hx, 0) + div_face(Fy, hy, 1)


def ch3d_rhs_fd(u, hx, hy, hz, kappa, mu_h, D):
    """3-D Cahn-Hilliard ``div(D(u) grad mu)``, cahn_hilliard.py:180-200 (7-point Laplacian
    derivatives.py:15-21; the face operators of axis 2 are :34-36,49-51,64-66)."""
    h = (hx, hy, hz)
    lap = sum((nb(u, 1, ax) - 2 * u + nb(u, -1, ax)) / h[ax] ** 2 for ax in range(3))
    mu = mu_h(u) - kappa * lap
    Du = D(u)
    out = 0.0
    for ax in range(3):
        out = out + div_face(avg_face(Du, ax) * grad_face(mu, h[ax], ax), h[ax], ax)
    return out


def ch3d_rhs_fourier(u, hx, hy, hz, kappa, mu_h, D):
    """Pseudo-spectral 3-D CH RHS (9 FFTs), cahn_hilliard.py:167-175; wave numbers as
    domains.py:44-47,58-60 with ``indexing='ij'``."""
    nx, ny, nz = u.shape
    kx, ky, kz = np.meshgrid(np.fft.fftfreq(nx, hx), np.fft.fftfreq(ny, hy), np.fft.fftfreq(nz, hz), indexing="ij")
    ik = [2j * np.pi * k for k in (kx, ky, kz)]
    k2 = ik[0] ** 2 + ik[1] ** 2 + ik[2] ** 2
    t_hat = np.fft.fftn(mu_h(u)) - kappa * k2 * np.fft.fftn(u)
    Du = D(u)
    acc = 0
    for a in ik:
        acc = acc + a * np.fft.fftn(Du * np.fft.ifftn(a * t_hat))
    return np.fft.ifftn(acc).real


def ac_rhs_fd(u, hx, hy, kappa, mu_h, R):
    """Allen-Cahn ``-R(u) mu``, allen_cahn.py:81-84."""
    return -R(u) * chem_potential(u, hx, hy, kappa, mu_h)


def sbm_norm_grad(psi, hx, hy):
    """|grad_c psi| / psi with periodic centred differences, allen_cahn.py:128-133."""
    gx = 0.5 * (nb(psi, 1, 0) - nb(psi, -1, 0)) / hx
    gy = 0.5 * (nb(psi, 1, 1) - nb(psi, -1, 1)) / hy
    return np.sqrt(gx**2 + gy**2) / psi


def sbm_inner(u, psi, hx, hy, kappa, f, mu_h, wall_weight):
    """mu_h(u) - kappa/psi div(psi grad u) - sqrt(kappa) |grad psi|/psi sqrt(2 f(u)) w,
    allen_cahn.py:139-155 / cahn_hilliard.py:257-276.  ``wall_weight`` is the field
    cos(theta) left_half [+ cos(pi - theta)(1 - left_half)]."""
    lap = div_face(avg_face(psi, 0) * grad_face(u, hx, 0), hx, 0) + div_face(
        avg_face(psi, 1) * grad_face(u, hy, 1), hy, 1
    )
    return (
        mu_h(u)
        - (kappa / psi) * lap
        - np.sqrt(kappa) * sbm_norm_grad(psi, hx, hy) * np.sqrt(2.0 * f(u)) * wall_weight
    )


def ac_sbm_rhs(u, psi, hx, hy, kappa, f, mu_h, R, theta_t, left_half):
    """Smoothed-boundary Allen-Cahn, allen_cahn.py:139-156."""
    return -R(u) * sbm_inner(u, psi, hx, hy, kappa, f, mu_h, np.cos(theta_t) * left_half)


def ch_sbm_rhs(u, psi, hx, hy, kappa, f, mu_h, D, theta_t, flux_t, left_half):
    """Smoothed-boundary Cahn-Hilliard, cahn_hilliard.py:257-289."""
    w = np.cos(theta_t) * left_half + np.cos(np.pi - theta_t) * (1.0 - left_half)
    inner = sbm_inner(u, psi, hx, hy, kappa, f, mu_h, w)
    Du = D(u)
    Fx = avg_face(psi, 0) * avg_face(Du, 0) * grad_face(inner, hx, 0)
    Fy = avg_face(psi, 1) * avg_face(Du, 1) * grad_face(inner, hy, 1)
    return (div_face(Fx, hx, 0) + div_face(Fy, hy, 1)) / psi + sbm_norm_grad(psi, hx, hy) * flux_t


def shape_smooth_rhs(u, hx, hy, eps, curvature):
    """Right-hand side of ``Shape.smooth_shape`` (shapes.py:41-64): Allen-Cahn smoothing of a mask whose diffusion
    acts along the normal for ``curvature = 0`` (no curvature flow) and isotropically for ``curvature = 1``:
    ``2 (c lap u + (1 - c) u_nn) - W'(u) / eps`` with ``W'(u) = 18 / eps * u (1 - u)(1 - 2u)`` and ``u_nn`` the
    second derivative along ``grad u`` (``|grad u|^2 < 1e-7`` replaced by 1, :53)."""
    gx, gy = grad_c(u, hx, 0), grad_c(u, hy, 1)
    gxx, gyy, gxy = grad2_c(u, hx, 0), grad2_c(u, hy, 1), grad2xy_c(u, hx, hy)
    g2 = gx**2 + gy**2
    g2 = np.where(g2 < 1e-7, 1.0, g2)
    unn = (gxx * gx**2 + 2.0 * gxy * gx * gy + gyy * gy**2) / g2
    pot = 18.0 / eps * u * (1.0 - u) * (1.0 - 2.0 * u)
    return 2.0 * (curvature * (gxx + gyy) + (1.0 - curvature) * unn) - pot / eps


def fft_wavenumbers(nx, ny, hx, hy):
    """``(2 pi i kx, 2 pi i ky)`` meshes, domains.py:44-47,58-60 and
    cahn_hilliard.py:65-67 (``fftfreq`` is cycles/unit; ``indexing='ij'``)."""
    kx, ky = np.meshgrid(np.fft.fftfreq(nx, hx), np.fft.fftfreq(ny, hy), indexing="ij")
    return 2j * np.pi * kx, 2j * np.pi * ky


def ch_fourier_symbol(nx, ny, hx, hy, kappa):
    """``kappa * ((2 pi i kx)^2 + (2 pi i ky)^2)^2``, cahn_hilliard.py:68-74."""
    ikx, iky = fft_wavenumbers(nx, ny, hx, hy)
    k2 = ikx**2 + iky**2
    return kappa * k2**2


def ch_rhs_fourier(u, hx, hy, kappa, mu_h, D):
    """Pseudo-spectral CH RHS (7 FFTs), cahn_hilliard.py:82-87."""
    ikx, iky = fft_wavenumbers(u.shape[0], u.shape[1], hx, hy)
    k2 = ikx**2 + iky**2
    t_hat = np.fft.fftn(mu_h(u)) - kappa * k2 * np.fft.fftn(u)
    Du = D(u)
    fx = np.fft.fftn(Du * np.fft.ifftn(ikx * t_hat))
    fy = np.fft.fftn(Du * np.fft.ifftn(iky * t_hat))
    return np.fft.ifftn(ikx * fx + iky * fy).real


def ac_rhs_fourier(u, hx, hy, kappa, mu_h, R):
    """Pseudo-spectral AC RHS, allen_cahn.py:74-79."""
    ikx, iky = fft_wavenumbers(u.shape[0], u.shape[1], hx, hy)
    k2 = ikx**2 + iky**2
    mu = np.fft.ifftn(np.fft.fftn(mu_h(u)) - kappa * k2 * np.fft.fftn(u)).real
    return -R(u) * mu


def ad_rhs_fd(u, hx, hy, vx_face, vy_face, diff):
    """Advection-diffusion ``-div(v u) + D lap(u)`` in conservative flux form.

    NOT in the reference package at this commit (only stale notebook call
    sites, notebooks/run_advection_diffusion.ipynb:67-72) -> parity unpinned.
    Defined with the reference's face primitives: ``vx_face[i,j]`` is the x
    velocity on face (i+1/2, j), ``vy_face`` the y velocity on face (i, j+1/2);
    the advected value on a face is the centre average (derivatives.py:39-46),
    the divergence is derivatives.py:54-61, diffusion is derivatives.py:8-12.
    """
    Fx = vx_face * avg_face(u, 0)
    Fy = vy_face * avg_face(u, 1)
    return -(div_face(Fx, hx, 0) + div_face(Fy, hy, 1)) + diff * lap5(u, hx, hy)


def gpe_b_terms(state, xmesh, ymesh, k, e, trap_factor, lights_field):
    """GPE pointwise operator B, gross_pitaevskii.py:67-75.

    ``state`` is ``(N, N, 2)`` real/imag; ``lights_field`` is ``lights(t, X, Y)``
    already evaluated.  Returns ``(N, N, 2)``.
    """
    psi = state[..., 0] + 1j * state[..., 1]
    b = (
        -0.5j * trap_factor * ((1 + e) * xmesh**2 + (1 - e) * ymesh**2)
        - 1j * lights_field
        - k * 1j * (np.abs(psi) ** 2)
    )
    return np.stack([b.real, b.imag], axis=-1)


# --------------------------------------------------------------------------
# one-step integrators
# --------------------------------------------------------------------------


def euler_step(f, t, y, dt):
    """Explicit Euler (diffrax.Euler; call site pde_env.py:293-303)."""
    return y + dt * f(t, y)


def rk4_step(f, t, y, dt):
    """Classical RK4.  Not in the reference; this is
    the textbook tableau wrapped around the reference RHS, written in the
    same accumulate-as-you-go order the HIP stage kernels use."""
    k = f(t, y)
    acc = y + (dt / 6) * k
    k = f(t + dt / 2, y + (dt / 2) * k)
    acc = acc + (dt / 3) * k
    k = f(t + dt / 2, y + (dt / 2) * k)
    acc = acc + (dt / 3) * k
    k = f(t + dt, y + dt * k)
    return acc + (dt / 6) * k


def imex_step(rhs, t0, y0, dt, A, symbol):
    """Semi-implicit Fourier-spectral step, numerics/solvers.py:56-63.
    (The unused ``euler_y1``/``y_error`` of :61,:65 are omitted.)"""
    f0 = rhs(t0, y0)
    denom = 1.0 + A * dt * symbol
    return y0 + dt * np.fft.ifftn(np.fft.fftn(f0) / denom).real


def strang_step(b_terms, t0, y0, dt, A_term, dx, time_scale):
    """Strang splitting step, numerics/solvers.py:99-115.

    ``b_terms(t, y)`` returns ``(N, N, 2)``; it is evaluated on the state
    *before* the first kinetic half step (:109), and the wavefunction is
    renormalised every step (:111), exactly as the reference does.
    """
    tau = dt * time_scale
    psi = y0[..., 0] + 1j * y0[..., 1]
    e_half = np.exp(A_term * 0.5 * tau)
    psi = np.fft.ifftn(np.fft.fftn(psi) * e_half)
    b = b_terms(t0, y0)
    psi = psi * np.exp((b[..., 0] + 1j * b[..., 1]) * tau)
    psi = psi / np.sqrt(np.sum(np.abs(psi) ** 2) * dx**2)
    psi = np.fft.ifftn(np.fft.fftn(psi) * e_half)
    return np.stack([psi.real, psi.imag], axis=-1)


def detect_vortices(psi, amp_thresh=0.0, tol=0.5):
    """Phase-circulation vortex census, pde_opt/rl_utils.py:19-84 (winding map and the three counts)."""
    two_pi = 2.0 * np.pi

    def wrap(x):  # rl_utils.py:14-16
        return (x + np.pi) % two_pi - np.pi

    theta = np.angle(psi)
    dth_x = wrap(nb(theta, 1, 1) - theta)
    dth_y = wrap(nb(theta, 1, 0) - theta)
    circulation = dth_x + nb(dth_y, 1, 1) - nb(dth_x, 1, 0) - dth_y
    n_float = circulation / two_pi
    n_int = np.rint(n_float).astype(np.int32)
    n_int = np.where(np.abs(n_float) >= tol, n_int, 0)
    if amp_thresh > 0.0:
        rho = np.abs(psi) ** 2
        rho_cell = 0.25 * (rho + nb(rho, 1, 0) + nb(rho, 1, 1) + nb(nb(rho, 1, 0), 1, 1))
        n_int = np.where(rho_cell >= amp_thresh, n_int, 0)
    return n_int, int((n_int != 0).sum()), int(n_int.sum()), int(np.abs(n_int).sum())


# Tsitouras 5(4) tableau (Ch. Tsitouras, Comput. Math. Appl. 62 (2011) 770-775).
# diffrax.Tsit5 (third party) uses these published coefficients.
_TS_C = (0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0)
_TS_A = (
    (0.161,),
    (-0.008480655492356989, 0.335480655492357),
    (2.8971530571054935, -6.359448489975075, 4.3622954328695815),
    (5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525),
    (5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
     -0.028269050394068383),
    (0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742,
     -3.290069515436081, 2.324710524099774),
)
_TS_B = _TS_A[5] + (0.0,)
# b - b_hat (error coefficients)
_TS_E = (
    0.00178001105222577714, 0.0008164344596567469, -0.007880878010261995,
    0.1447110071732629, -0.5823571654525552, 0.45808210592918697, -1.0 / 66.0,
)


def tsit5_dense_weights(theta):
    """b_i(theta) of the 4th-order continuous extension of the pair (Tsitouras 2011, section 4) -- the
    interpolant diffrax.Tsit5 (third party) evaluates at SaveAt points inside a step; b_i(1) = _TS_B."""
    th, t2 = theta, theta * theta
    return (
        -1.0530884977290216 * th * (th - 1.3299890189751412) * (t2 - 1.4364028541716351 * th + 0.7139816917074209),
        0.1017 * t2 * (t2 - 2.1966568338249754 * th + 1.2949852507374631),
        2.490627285651252793 * t2 * (t2 - 2.38535645472061657 * th + 1.57803468208092486),
        -16.54810288924490272 * (th - 1.21712927295533244) * (th - 0.61620406037800089) * t2,
        47.37952196281928122 * (th - 1.203071208372362603) * (th - 0.658047292653547382) * t2,
        -34.87065786149660974 * (th - 1.2) * (th - 0.666666666666666667) * t2,
        2.5 * (th - 1.0) * (th - 0.6) * t2,
    )


def tsit5_dense(y, dt, ks, theta):
    """``y(t + theta dt) = y + dt sum_i b_i(theta) k_i`` from the seven slopes of a step"""
    out = y
    for b, kk in zip(tsit5_dense_weights(theta), ks):
        out = out + (dt * b) * kk
    return out


def tsit5_step(f, t, y, dt, k1=None, return_slopes=False):
    """One Tsit5 step.  Returns ``(y1, err, k7)`` (FSAL: ``k7 = f(t+dt, y1)``); with ``return_slopes`` the
    seven slopes are appended (dense output)."""
    ks = [f(t, y) if k1 is None else k1]
    for s in range(6):
        ys = y
        for a, kk in zip(_TS_A[s], ks):
            ys = ys + (dt * a) * kk
        ks.append(f(t + _TS_C[s] * dt, ys))
        if s == 5:
            y1 = ys
    err = 0
    for e, kk in zip(_TS_E, ks):
        err = err + (dt * e) * kk
    if return_slopes:
        return y1, err, ks[6], ks
    return y1, err, ks[6]


# --------------------------------------------------------------------------
# stepping loop (diffrax.diffeqsolve + ConstantStepSize + SaveAt)
# --------------------------------------------------------------------------


def constant_step_plan(t0, t1, dt, rel_tol=1e-9):
    """``(n_full, remainder)``: the step sequence of a constant-step solve.

    diffrax's loop is ``while t < t1: t_next = min(t + dt, t1)`` (third party;
    call sites pde_env.py:293-303, pde_model.py:120-134): ``n_full`` steps of
    exactly ``dt`` and, if ``t1 - t0`` is not a multiple of ``dt``, one final
    clipped step.  Floating-point accumulation of ``t`` upstream can add a
    final step of O(eps) length; remainders below ``rel_tol * dt`` are dropped
    here (documented deviation; it changes the state by <= rel_tol*dt*|f|).
    """
    span = float(t1) - float(t0)
    if span <= 0:
        return 0, 0.0
    n_full = int(math.floor(span / dt + rel_tol))
    rem = span - n_full * dt
    if rem <= rel_tol * dt:
        rem = 0.0
    return n_full, rem


def integrate(step, y0, t0, t1, dt):
    """Advance ``y0`` from ``t0`` to ``t1`` with ``step(t, y, dt) -> y``."""
    n_full, rem = constant_step_plan(t0, t1, dt)
    y, t = y0, float(t0)
    for i in range(n_full):
        y = step(t0 + i * dt, y, dt)
    if rem > 0.0:
        y = step(t0 + n_full * dt, y, rem)
    return y


def solve_saveat(step, y0, ts, dt):
    """Constant-step solve with ``SaveAt(ts=ts)`` and linear dense output
    (``LocalLinearInterpolation``, numerics/solvers.py:48,91; pde_model.py:128).

    Steps do not stop at save points; a save point inside ``(t_n, t_{n+1}]`` is
    the linear interpolant of the two step end points.
    """
    ts = [float(t) for t in ts]
    t0, t1 = ts[0], ts[-1]
    n_full, rem = constant_step_plan(t0, t1, dt)
    edges = [t0 + i * dt for i in range(n_full + 1)]
    if rem > 0.0:
        edges.append(t1)
    edges[-1] = t1
    out = []
    y_prev, y = y0, y0
    step_idx = 0  # y is the state at edges[step_idx]
    for tq in ts:
        while step_idx < len(edges) - 1 and edges[step_idx] < tq:
            y_prev = y
            y = step(edges[step_idx], y, edges[step_idx + 1] - edges[step_idx])
            step_idx += 1
        if step_idx == 0 or edges[step_idx] <= tq:
            out.append(y)
        else:
            a, b = edges[step_idx - 1], edges[step_idx]
            th = (tq - a) / (b - a)
            out.append(y_prev + th * (y - y_prev))
    return np.stack(out)


# --------------------------------------------------------------------------
# Legendre closures (pde_opt/numerics/functions/legendre.py)
# --------------------------------------------------------------------------


def legendre_series(coeffs, x):
    """sum_n c_n P_n(x) by the three-term recurrence, legendre.py:23-34."""
    coeffs = list(coeffs)
    res = coeffs[0] * np.ones_like(x)
    if len(coeffs) > 1:
        res = res + coeffs[1] * x
    p_prev, p_cur = np.ones_like(x), x
    for n in range(2, len(coeffs)):
        p_next = ((2 * n - 1) * x * p_cur - (n - 1) * p_prev) / n
        res = res + coeffs[n] * p_next
        p_prev, p_cur = p_cur, p_next
    return res


def diffusion_legendre(coeffs, c):
    """exp(Legendre(2c-1)), legendre.py:48-53."""
    return np.exp(legendre_series(coeffs, 2.0 * c - 1.0))


def chem_potential_legendre(coeffs, c, prior=None):
    """Legendre(2c-1) [+ prior(c)], legendre.py:67-74."""
    r = legendre_series(coeffs, 2.0 * c - 1.0)
    return r if prior is None else r + prior(c)
