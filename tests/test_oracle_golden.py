"""The CPU oracle (oracle/np_oracle.py) against goldens produced by the reference's
own source (oracle/gen_golden.py) -- this is what pins the oracle."""

import numpy as np
import pytest

from oracle import np_oracle as O

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


def _cases(z, kind):
    keys = sorted(k[: -len("/rhs")] for k in z.files if k.startswith(kind + "/") and k.endswith("/rhs"))
    return keys


def _spacing(tag):
    shape, dt = tag.split("_")
    nx, ny = (int(v) for v in shape.split("x"))
    # Domain of gen_golden: L = 0.01*N  ->  dx = (hi-lo)/N  (domains.py:30-33)
    lx, ly = 0.01 * nx, 0.01 * ny
    return (lx / 2 - (-lx / 2)) / nx, (ly / 2 - (-ly / 2)) / ny


@pytest.mark.parametrize("kind", ["ch_fd", "ac_fd", "ch_fourier", "ac_fourier"])
def test_rhs_matches_reference_goldens(golden, kind):
    z = golden("rhs_cases.npz")
    fn = {"ch_fd": O.ch_rhs_fd, "ac_fd": O.ac_rhs_fd, "ch_fourier": O.ch_rhs_fourier, "ac_fourier": O.ac_rhs_fourier}[kind]
    keys = _cases(z, kind)
    assert keys
    for key in keys:
        _, mu, mob, tag = key.split("/")
        u, want = z[key + "/u"], z[key + "/rhs"]
        hx, hy = _spacing(tag)
        got = fn(u, hx, hy, 0.002, MU[mu], MOB[mob])
        assert got.dtype == want.dtype, key
        if kind.endswith("fd"):
            # same operations in the same order -> bitwise
            np.testing.assert_array_equal(got, want, err_msg=key)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max(), err_msg=key)


def test_fourier_symbol(golden):
    z = golden("rhs_cases.npz")
    for k in z.files:
        if k.endswith("/symbol"):
            tag = k.split("/")[3]
            nx, ny = (int(v) for v in tag.split("_")[0].split("x"))
            hx, hy = _spacing(tag)
            np.testing.assert_array_equal(O.ch_fourier_symbol(nx, ny, hx, hy, 0.002), z[k])


def test_manufactured_solution_slope(golden):
    """tests/test_rhs_convergence.py:14-77 -- slope 2.0 +- 10 % -- run on the oracle."""
    import sympy as sp
    from sympy.utilities.lambdify import lambdify

    x, y, t = sp.symbols("x y t", real=True)
    u = sp.sin(2 * x) * sp.cos(3 * y) * sp.exp(-0.7 * t)
    kappa = 1e-2
    mu = u**3 - u - kappa * (sp.diff(u, x, 2) + sp.diff(u, y, 2))
    D = 1 + u**2
    ch = sp.diff(D * sp.diff(mu, x), x) + sp.diff(D * sp.diff(mu, y), y)
    ac = -D * mu
    u_fn = lambdify((x, y, t), u, "numpy")
    z = golden("manufactured.npz")
    for name, expr, fn in (("ch", ch, O.ch_rhs_fd), ("ac", ac, O.ac_rhs_fd)):
        ex_fn = lambdify((x, y, t), expr, "numpy")
        hs, errs = [], []
        for n in (32, 64, 128, 256):
            L = 2 * np.pi
            h = L / n
            ax = np.linspace(-L / 2 + h / 2, L / 2 - h / 2, n)
            X, Y = np.meshgrid(ax, ax, indexing="ij")
            ue = u_fn(X, Y, 0.0)
            got = fn(ue, h, h, kappa, MU["cubic"], MOB["one_plus_sq"])
            exact = ex_fn(X, Y, 0.0)
            if f"{name}/{n}/rhs_fd" in z.files:
                np.testing.assert_allclose(got, z[f"{name}/{n}/rhs_fd"], rtol=0, atol=1e-12 * np.abs(exact).max())
                np.testing.assert_allclose(exact, z[f"{name}/{n}/rhs_exact"], rtol=1e-9, atol=1e-9)
            errs.append(np.sqrt(np.sum((got - exact) ** 2)) / np.sqrt(np.sum(exact**2)))
            hs.append(h)
        slope = np.polyfit(np.log(hs), np.log(errs), 1)[0]
        np.testing.assert_allclose(slope, 2.0, rtol=0.1)


def test_imex_trajectory(golden):
    z = golden("trajectories.npz")
    y = z["imex/y0"]
    h = 0.64 / 64
    sym = O.ch_fourier_symbol(64, 64, h, h, 0.002)
    rhs = lambda t, u: O.ch_rhs_fd(u, h, h, 0.002, MU["regsol"], MOB["c1mc"])
    dt = float(z["imex/dt"])
    for i in range(10):
        y = O.imex_step(rhs, i * dt, y, dt, 0.5, sym)
        np.testing.assert_allclose(y, z["imex/ys"][i], rtol=0, atol=1e-14)


def test_imex_1d(golden):
    z = golden("trajectories.npz")
    y = z["imex1d/y0"]
    hx, hy = 2.56 / 256, 0.01
    sym = O.ch_fourier_symbol(256, 1, hx, hy, 0.002)
    rhs = lambda t, u: O.ch_rhs_fd(u, hx, hy, 0.002, MU["cubic"], MOB["one"])
    y = O.integrate(lambda t, u, dt: O.imex_step(rhs, t, u, dt, 0.5, sym), y, 0.0, 200 * 5e-5, 5e-5)
    np.testing.assert_allclose(y, z["imex1d/y200"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name,tscale", [("zeroA_imag", -1j), ("realA_real", 1.0), ("realA_imag", -1j)])
def test_strang_trajectory(golden, name, tscale):
    z = golden("trajectories.npz")
    n = 48
    h = 24.0 / n
    ax = np.linspace(-12 + h / 2, 12 - h / 2, n)
    X, Y = np.meshgrid(ax, ax, indexing="ij")
    lights = 0.05 * X
    b = lambda t, yy: O.gpe_b_terms(yy, X, Y, 1000.0, 0.1, 1.0, lights)
    np.testing.assert_allclose(b(0.0, z["strang/y0"]), z["strang/b_terms"], rtol=1e-14, atol=1e-14)
    ikx, iky = O.fft_wavenumbers(n, n, h, h)
    a_term = 0.5j * (ikx**2 + iky**2) * (0.0 if name.startswith("zeroA") else 1.0)
    y = z["strang/y0"]
    for i in range(5):
        y = O.strang_step(b, i * 1e-3, y, 1e-3, a_term, h, tscale)
        np.testing.assert_allclose(y, z[f"strang/{name}/ys"][i], rtol=0, atol=1e-13)


def test_domain_golden(golden):
    z = golden("domain_8x6.npz")
    h0, h1 = 2.0 / 8, 3.0 / 6
    np.testing.assert_array_equal(z["dx"], [h0, h1])
    np.testing.assert_allclose(z["ax0"], np.linspace(-1 + h0 / 2, 1 - h0 / 2, 8), rtol=0, atol=0)
    np.testing.assert_array_equal(z["f1"], np.fft.fftfreq(6, h1))


def test_tsit5_tableau_consistency():
    # row sums equal the nodes; 5th-order weights integrate polynomials exactly
    for c, row in zip(O._TS_C, O._TS_A):
        assert abs(sum(row) - c) < 1e-14
    b = np.array(O._TS_B)
    c = np.array((0.0,) + O._TS_C)
    for p in range(5):
        assert abs(np.sum(b * c**p) - 1 / (p + 1)) < 1e-13
    assert abs(sum(O._TS_E)) < 1e-15
    # one step on y' = -y is 5th-order accurate
    y1, err, _ = O.tsit5_step(lambda t, y: -y, 0.0, np.array([1.0]), 0.1)
    assert abs(y1[0] - np.exp(-0.1)) < 1e-8
    assert abs(err[0]) < 1e-5


def test_rk4_order():
    f = lambda t, y: -y
    errs = []
    for n in (10, 20, 40):
        y = O.integrate(lambda t, y, dt: O.rk4_step(f, t, y, dt), np.array([1.0]), 0.0, 1.0, 1.0 / n)
        errs.append(abs(y[0] - np.exp(-1)))
    assert 3.8 < np.log2(errs[0] / errs[1]) < 4.2


def test_constant_step_plan():
    assert O.constant_step_plan(0.0, 0.05, 1e-4) == (500, 0.0)
    n, rem = O.constant_step_plan(0.0, 1.0, 0.3)
    assert n == 3 and abs(rem - 0.1) < 1e-12
    ys = O.solve_saveat(lambda t, y, dt: y + dt, np.zeros(1), [0.0, 0.25, 0.5, 1.0], 0.1)
    np.testing.assert_allclose(ys[:, 0], [0.0, 0.25, 0.5, 1.0], atol=1e-12)


# ---- smoothed-boundary equations ---------------------------------------
SBM_F = lambda c: c * np.log(c) + (1.0 - c) * np.log(1.0 - c) + 3.0 * c * (1.0 - c) + 0.059  # noqa: E731
SBM_THETA = lambda t: 34.9065850398866 * t**2 - 10.4719755119660 * t + np.pi / 2  # noqa: E731
SBM_FLUX = lambda t: 0.02 * (1.0 + 3.0 * t)  # noqa: E731


def sbm_cases(z):
    return sorted(k[: -len("/psi")] for k in z.files if k.endswith("/psi"))


def test_sbm_rhs_matches_reference_goldens(golden):
    z = golden("sbm_cases.npz")
    keys = sbm_cases(z)
    assert len(keys) == 8
    for key in keys:
        kind = key.split("/")[0]
        psi, u, lh = z[key + "/psi"], z[key + "/u"], z[key + "/left_half"]
        np.testing.assert_allclose(O.sbm_norm_grad(psi, 1.0, 1.0), z[key + "/norm_grad_psi"], rtol=1e-6)
        for t in (0.0, 0.17):
            want = z[f"{key}/rhs_t{t}"]
            if kind == "ac":
                got = O.ac_sbm_rhs(u, psi, 1.0, 1.0, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA(t), lh)
            else:
                got = O.ch_sbm_rhs(u, psi, 1.0, 1.0, 1.5, SBM_F, MU["regsol"], MOB["c1mc"], SBM_THETA(t),
                                   SBM_FLUX(t), lh)
            tol = 1e-12 if want.dtype == np.float64 else 2e-5
            np.testing.assert_allclose(got, want, rtol=0, atol=tol * np.abs(want).max(), err_msg=f"{key} t={t}")


def test_detect_vortices_matches_reference(golden):
    """oracle restatement of rl_utils.detect_vortices (pde_opt/rl_utils.py:19-84) vs the reference's output"""
    z = golden("vortices.npz")
    tags = sorted({k.split("/")[0] for k in z.files})
    assert len(tags) == 2
    for tag in tags:
        psi = z[f"{tag}/psi"]
        for amp, tol in ((0.0, 0.5), (0.02, 0.5), (0.0, 1.5)):
            key = f"{tag}/amp{amp}_tol{tol}"
            w, num, total, abs_c = O.detect_vortices(psi, amp, tol)
            np.testing.assert_array_equal(w, z[key + "/winding"])
            np.testing.assert_array_equal([num, total, abs_c], z[key + "/counts"])
    assert z["48x48_c128/amp0.0_tol0.5/counts"][0] > 0


def _dx3(tag):
    nx, ny, nz = (int(v) for v in tag.split("_")[0].split("x"))
    # Domain of gen_golden: box (-0.005 nx, 0.005 nx) x (-0.005 ny, 0.005 ny) x (0, 0.012 nz)
    return (0.01 * nx) / nx, (0.01 * ny) / ny, (0.012 * nz) / nz


def test_ch3d_rhs_matches_reference_goldens(golden):
    """CahnHilliard3DPeriodic.rhs_fd (cahn_hilliard.py:180-200): same operations, same order -> bitwise"""
    z = golden("ch3d_cases.npz")
    tags = sorted(k[: -len("/rhs")] for k in z.files if k.endswith("/rhs"))
    assert len(tags) == 6
    for tag in tags:
        hx, hy, hz = _dx3(tag)
        got = O.ch3d_rhs_fd(z[tag + "/u"], hx, hy, hz, 0.002, MU["regsol"], MOB["c1mc"])
        np.testing.assert_array_equal(got, z[tag + "/rhs"], err_msg=tag)


def test_advection_diffusion_manufactured_solution_slope():
    """advection-diffusion is absent from the reference package, so its pin is a
    sympy manufactured solution in the style of tests/test_rhs_convergence.py:14-77: second-order slope
    (2.0 +- 10 %) of the oracle's conservative flux form against the exact -div(v u) + D lap u."""
    import sympy as sp
    from sympy.utilities.lambdify import lambdify

    x, y, t = sp.symbols("x y t", real=True)
    u = sp.sin(2 * x) * sp.cos(3 * y) * sp.exp(-0.7 * t)
    vx = sp.Rational(3, 5) + sp.Rational(3, 10) * sp.sin(x) * sp.cos(2 * y) * (1 + t)
    vy = -sp.Rational(2, 5) + sp.Rational(1, 5) * sp.cos(3 * x) * sp.sin(y)
    Dc = 0.05
    exact = -(sp.diff(vx * u, x) + sp.diff(vy * u, y)) + Dc * (sp.diff(u, x, 2) + sp.diff(u, y, 2))
    u_fn, ex_fn = lambdify((x, y, t), u, "numpy"), lambdify((x, y, t), exact, "numpy")
    vx_fn, vy_fn = lambdify((x, y, t), vx, "numpy"), lambdify((x, y, t), vy, "numpy")
    hs, errs = [], []
    for n in (32, 64, 128, 256):
        h = 2 * np.pi / n
        ax = np.linspace(-np.pi + h / 2, np.pi - h / 2, n)
        X, Y = np.meshgrid(ax, ax, indexing="ij")
        got = O.ad_rhs_fd(u_fn(X, Y, 0.3), h, h, vx_fn(X + h / 2, Y, 0.3), vy_fn(X, Y + h / 2, 0.3) + 0 * X, Dc)
        ex = ex_fn(X, Y, 0.3)
        errs.append(np.sqrt(np.sum((got - ex) ** 2)) / np.sqrt(np.sum(ex**2)))
        hs.append(h)
    slope = np.polyfit(np.log(hs), np.log(errs), 1)[0]
    np.testing.assert_allclose(slope, 2.0, rtol=0.1)


def _shape_cases(z):
    return sorted(k[: -len("/rhs")] for k in z.files if k.endswith("/rhs"))


def test_shape_smoothing_rhs_matches_reference_goldens(golden):
    """``Shape.smooth_shape``'s right-hand side (shapes.py:44-64), evaluated by the reference's own closure in
    oracle/gen_golden.py: smeared fields and the raw masks (where |grad u|^2 < 1e-7 takes the :53 branch)"""
    z = golden("shapes.npz")
    keys = _shape_cases(z)
    assert len(keys) == 6
    for key in keys:
        par = key.split("/")[1].split("_")
        hx, hy, eps, c = float(par[0][2:]), float(par[1]), float(par[2][3:]), float(par[3][1:])
        for field, out in (("u", "rhs"), (None, "rhs_binary")):
            u = z[key + "/u"] if field else z[key.split("/")[0] + "/mask"]
            got = O.shape_smooth_rhs(u, hx, hy, eps, c)
            np.testing.assert_allclose(got, z[f"{key}/{out}"], rtol=0, atol=1e-13 * np.max(np.abs(z[f"{key}/{out}"])))
