"""Closures outside the in-kernel family (SURVEY Appendix D's escape hatch; the reference takes ANY pointwise callable:
cahn_hilliard.py:51-54, allen_cahn.py:47-50, functions/legendre.py:56-74): traced with sympy, vetted node by node,
emitted as C function bodies and compiled at run time into the generic stage kernel (csrc/jit.hip, jit_device.hpp).
Here, without a GPU: the tracer / emitter, and that every emitted body COMPILES for gfx950 (hiprtc needs no device) --
including the layout check of the kernel's StageArgs copy against the library's.  The numerics are tests/test_gpu_jit.py."""
import ctypes as C

import numpy as np
import pytest

import pde_opt_amd as P
from pde_opt_amd import _lib as L
from pde_opt_amd.numerics.closures import JIT, as_closure, jit_body_of

OUTSIDE = {
    "tanh": lambda c: np.tanh(3 * c),
    "sqrt-exp": lambda c: np.sqrt(c) * (1 - c) + np.exp(-c**2),
    "real-power": lambda c: c**2.5 + 1 / (1 + c**2),
    "log-mix": lambda c: np.log(1 + c**2) - 0.3 * np.tanh(c - 0.5) ** 2,
}


def _compiles(mu_body, mob_body, dtype):
    lib = L.load_library()
    log = C.create_string_buffer(8000)
    rc = lib.pdeopt_jit_check(L.dtype_code(dtype), mu_body.encode(), mob_body.encode(), log, 8000)
    return rc, log.value.decode()


@pytest.mark.parametrize("name", sorted(OUTSIDE))
def test_callable_outside_the_family_is_traced_emitted_and_compiles(name):
    fn = OUTSIDE[name]
    d = as_closure(fn)
    assert d.kind == JIT and d.source.startswith("return ") and d.source.endswith(";") and "\n" not in d.source
    x = np.linspace(0.05, 0.95, 11)
    np.testing.assert_allclose(d(x), fn(x), rtol=1e-14, atol=1e-15)  # the host evaluator is the traced expression
    # every numeric literal is typed (a bare 3.0 would promote an fp32 kernel's arithmetic to double)
    import re

    assert not re.search(r"(?<![\w(])\d+\.\d+", d.source.replace("T(", "(")) or all(
        tok.startswith("T(") for tok in re.findall(r"T\([^()]*\)", d.source))
    mob = jit_body_of(as_closure(lambda c: c * (1 - c)))  # a family member spelled out: both roles are compiled
    for dtype in (np.float32, np.float64):
        rc, log = _compiles(d.source, mob, dtype)
        assert rc == 0, log


def test_family_members_as_bodies_compile_and_are_the_family():
    """when one role is compiled the other is too (the compiled kernel has no table-driven evaluator): polynomial, logit
    prior, mixing entropy, exp-wrapped and Legendre members spelled out as statements"""
    descs = [as_closure(lambda c: c**3 - c), as_closure(lambda c: np.log(c / (1 - c)) + 3 * (1 - 2 * c)),
             as_closure(lambda c: np.exp(0.5 - c)), P.DiffusionLegendrePolynomials(np.array([0.1, -0.2, 0.05])).closure_desc(),
             P.ChemicalPotentialLegendrePolynomials(np.array([0.3, 0.1, -0.2, 0.05]), prior_fn=lambda v: np.log(v / (1 - v))).closure_desc()]
    for d in descs:
        assert d.kind != JIT
        body = jit_body_of(d)
        assert body.endswith("return r;") and "\n" not in body
        rc, log = _compiles(body, body, np.float64)
        assert rc == 0, (body, log)


def test_legendre_potential_with_an_arbitrary_pointwise_prior():
    """functions/legendre.py:56-74 accepts any callable prior: tanh(c) is outside the family -> series by its recurrence +
    the prior's expression, compiled"""
    ch = P.ChemicalPotentialLegendrePolynomials(np.array([0.3, 0.1, -0.2, 0.05]), prior_fn=lambda v: np.tanh(2 * v) + v**2)
    d = ch.closure_desc()
    assert d.kind == JIT and "tanh" in d.source and "pn" in d.source
    cc = np.linspace(0.05, 0.95, 9)
    np.testing.assert_allclose(d(cc), ch(cc), rtol=1e-13, atol=1e-14)
    rc, log = _compiles(d.source, jit_body_of(as_closure(0.15)), np.float32)
    assert rc == 0, log


def test_what_stays_refused():
    with pytest.raises(P.UnsupportedClosureError):
        as_closure(lambda c: np.sin(c))  # not in the vetted node set
    with pytest.raises(P.UnsupportedClosureError):
        as_closure(lambda c: np.roll(c, 1))  # non-pointwise (CNN / Mixer closures: out of scope)
    with pytest.raises(P.UnsupportedClosureError):
        P.ChemicalPotentialLegendrePolynomials(np.array([0.3, 0.1]), prior_fn=lambda v: np.sin(v)).closure_desc()


def test_a_body_that_does_not_compile_reports_the_compiler_message():
    rc, log = _compiles("return undefined_function(c);", "return T(1);", np.float32)
    assert rc != 0 and "undefined_function" in log
    lib = L.load_library()
    assert lib.pdeopt_jit_check(L.dtype_code(np.float32), None, b"return T(1);", None, 0) != 0
