"""The C restatement (oracle/c_oracle.c) against the reference goldens and the numpy oracle."""
import numpy as np

from oracle import c_oracle as CO
from oracle import np_oracle as O
from util import MOB, MU, rel_l2

CL = {
    "cubic": CO.closure(0, 0, (0.0, -1.0, 0.0, 1.0)),
    "regsol": CO.closure(0, 1, (3.0, -6.0)),
    "one": CO.closure(0, 0, (1.0,)),
    "c1mc": CO.closure(0, 0, (0.0, 1.0, -1.0)),
    "one_plus_sq": CO.closure(0, 0, (1.0, 0.0, 1.0)),
    "const015": CO.closure(0, 0, (0.15,)),
}


def test_c_oracle_matches_reference_goldens(golden):
    z = golden("rhs_cases.npz")
    n = 0
    for kind, eq in (("ch_fd", 0), ("ac_fd", 1)):
        for key in sorted(k[:-4] for k in z.files if k.startswith(kind + "/") and k.endswith("/rhs")):
            _, mu, mob, tag = key.split("/")
            nx, ny = (int(v) for v in tag.split("_")[0].split("x"))
            u, want = z[key + "/u"], z[key + "/rhs"]
            got = CO.rhs(eq, u, 0.01, 0.01, 0.002, CL[mu], CL[mob])
            # same formulas, different association (Horner closures, fused loops): rounding only
            tol = 1e-12 if want.dtype == np.float64 else 2e-5
            assert rel_l2(got, want) < tol, (key, rel_l2(got, want))
            n += 1
    assert n >= 40


def test_c_oracle_rk4_matches_numpy_oracle():
    rng = np.random.default_rng(0)
    y0 = np.clip(0.5 + 0.05 * rng.standard_normal((48, 40)), 0.05, 0.95)
    f = lambda t, u: O.ch_rhs_fd(u, 0.01, 0.01, 0.002, MU["regsol"], MOB["c1mc"])
    ref = y0
    for _ in range(6):
        ref = O.rk4_step(f, 0.0, ref, 2e-7)
    got = CO.rk4(0, y0, 0.01, 0.01, 0.002, CL["regsol"], CL["c1mc"], 2e-7, 6)
    assert rel_l2(got - y0, ref - y0) < 1e-11
    got32 = CO.rk4(0, y0.astype(np.float32), 0.01, 0.01, 0.002, CL["regsol"], CL["c1mc"], 2e-7, 6)
    assert np.max(np.abs(got32 - ref)) < 5e-7
