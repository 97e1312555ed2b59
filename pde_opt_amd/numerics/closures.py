"""Pointwise closures mu(u), D(u), R(u): from Python callables to the in-kernel family.

Upstream these are arbitrary Python callables traced by ``jax.jit`` into the RHS (dataclass
fields pde_opt/numerics/equations/cahn_hilliard.py:51-54, allen_cahn.py:47-50).  A HIP kernel
needs a closed family; the one implemented (include/pdeopt_hip.h) is

    f(c) = series(c) [+ log(c / (1 - c))] [+ c log c + (1 - c) log(1 - c)]  [then exp(.)]

with ``series`` a polynomial in ``c`` or a Legendre series in ``2c - 1``.  ``as_closure`` turns a
user callable into a descriptor by evaluating it once on a symbolic variable (sympy) and matching
the result against that family; anything else (CNN / MLP-Mixer closures are not pointwise) is
rejected with ``UnsupportedClosureError`` instead of silently running somewhere slower.
"""

from __future__ import annotations

import dataclasses
import numbers
from typing import Callable, Sequence

import numpy as np

POLY, LEGENDRE, JIT = 0, 1, 2
LOGIT_PRIOR, EXP_WRAP, MIX_ENTROPY = 1, 2, 4
MAX_COEF = 16


class UnsupportedClosureError(ValueError):
    pass


@dataclasses.dataclass(frozen=True)
class ClosureDesc:
    """A member of the in-kernel closure family.  Callable on numpy arrays (host utility only)."""

    kind: int = POLY
    flags: int = 0
    coef: tuple = (0.0,)
    # kind == JIT (a callable outside the family, compiled at run time: csrc/jit.hip): the C function body emitted from
    # the traced expression, and a host evaluator of the same expression
    source: str = ""
    host_fn: object = dataclasses.field(default=None, compare=False, repr=False)

    def __post_init__(self):
        if not 1 <= len(self.coef) <= MAX_COEF:
            raise UnsupportedClosureError(
                f"closure needs 1..{MAX_COEF} coefficients, got {len(self.coef)}"
            )

    def __call__(self, c):
        c = np.asarray(c)
        if self.kind == JIT:
            return self.host_fn(c)
        if self.kind == POLY:
            r = np.zeros_like(c) + self.coef[-1]
            for a in self.coef[-2::-1]:
                r = r * c + a
        else:
            r = np.polynomial.legendre.legval(2.0 * c - 1.0, np.asarray(self.coef))
        if self.flags & LOGIT_PRIOR:
            r = r + np.log(c / (1 - c))
        if self.flags & MIX_ENTROPY:
            r = r + c * np.log(c) + (1 - c) * np.log(1 - c)
        if self.flags & EXP_WRAP:
            r = np.exp(r)
        return r

    def with_coef(self, coef: Sequence[float]) -> "ClosureDesc":
        return dataclasses.replace(self, coef=tuple(float(v) for v in coef))


def polynomial(*coef: float, logit_prior: bool = False) -> ClosureDesc:
    """``sum_k coef[k] c^k`` (+ ``log(c/(1-c))``)."""
    return ClosureDesc(POLY, LOGIT_PRIOR if logit_prior else 0, tuple(float(v) for v in coef))


def constant(v: float) -> ClosureDesc:
    return polynomial(v)


# ----------------------------------------------------------------------------------------------
# symbolic tracing of user callables
# ----------------------------------------------------------------------------------------------


class _Sym:
    """Stand-in for an array during tracing: arithmetic and numpy ufuncs build a sympy expr."""

    __array_priority__ = 1000

    def __init__(self, expr):
        self.e = expr

    @staticmethod
    def _u(x):
        return x.e if isinstance(x, _Sym) else x

    def _bin(self, other, op):
        import sympy as sp

        o = self._u(other)
        if isinstance(o, np.ndarray):
            if o.ndim == 0:
                o = float(o)
            else:
                raise UnsupportedClosureError("closures may only combine the field with scalars")
        if isinstance(o, numbers.Real) and not isinstance(o, bool):
            o = sp.Rational(float(o))  # exact value of the double: coefficients round-trip
        return _Sym(op(self.e, o))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: b / a)
    def __pow__(self, o): return self._bin(o, lambda a, b: a**b)
    def __neg__(self): return _Sym(-self.e)
    def __pos__(self): return self

    # numpy protocol: np.log(sym), np.exp(sym), np.ones_like(sym), ...
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        import sympy as sp

        if method != "__call__":
            raise UnsupportedClosureError(f"numpy ufunc method {method} is not traceable")
        name = ufunc.__name__
        a = [self._u(x) for x in inputs]
        table = {
            "log": lambda x: sp.log(x), "exp": lambda x: sp.exp(x), "negative": lambda x: -x,
            "square": lambda x: x**2, "sqrt": lambda x: sp.sqrt(x), "positive": lambda x: x,
            "add": lambda x, y: x + y, "subtract": lambda x, y: x - y, "multiply": lambda x, y: x * y,
            "divide": lambda x, y: x / y, "true_divide": lambda x, y: x / y, "power": lambda x, y: x**y,
            "reciprocal": lambda x: 1 / x, "tanh": lambda x: sp.tanh(x),
        }
        if name not in table:
            raise UnsupportedClosureError(f"numpy.{name} is outside the supported closure family")
        return _Sym(table[name](*a))

    def __array_function__(self, func, types, args, kwargs):
        import sympy as sp

        if func is np.ones_like:
            return _Sym(sp.Integer(1))
        if func is np.zeros_like:
            return _Sym(sp.Integer(0))
        if func is np.full_like:
            return _Sym(sp.Rational(float(args[1])))
        raise UnsupportedClosureError(f"numpy.{func.__name__} is outside the supported closure family")

    # jnp-style helpers commonly used inside closures
    def log(self): return np.log(self)
    def exp(self): return np.exp(self)


def _match(expr, c):
    import sympy as sp

    flags = 0
    e = sp.expand(sp.expand_log(sp.sympify(expr), force=True))
    if e.has(sp.exp):
        # exp(series): log of the expression must collapse to a polynomial
        inner = sp.expand(sp.expand_log(sp.log(e), force=True))
        if inner.is_polynomial(c) and not inner.has(sp.log) and not inner.has(sp.exp):
            flags |= EXP_WRAP
            e = inner
    logit = sp.log(c) - sp.log(1 - c)
    entropy = sp.expand(c * sp.log(c) + (1 - c) * sp.log(1 - c))
    for candidate_flags, cand in (
        (0, e),
        (LOGIT_PRIOR, sp.expand(e - logit)),
        (MIX_ENTROPY, sp.expand(e - entropy)),
        (LOGIT_PRIOR | MIX_ENTROPY, sp.expand(e - logit - entropy)),
    ):
        # logs written as log(1 - c) or log(-(c - 1)) etc. must cancel exactly to count
        if cand.is_polynomial(c) and not cand.has(sp.log):
            poly = sp.Poly(cand, c)
            coef = [float(v) for v in reversed(poly.all_coeffs())]
            if len(coef) > MAX_COEF:
                raise UnsupportedClosureError(
                    f"polynomial degree {len(coef) - 1} exceeds the in-kernel limit {MAX_COEF - 1}"
                )
            return ClosureDesc(POLY, flags | candidate_flags, tuple(coef))
    # outside the family: a pointwise expression of a vetted node set is compiled at run time (csrc/jit.hip)
    return jit_closure(sp.sympify(expr), c)


# ----------------------------------------------------------------------------------------------
# closures outside the family: C function bodies for run-time compilation (csrc/jit.hip, jit_device.hpp)
# ----------------------------------------------------------------------------------------------
def _emit_c(e, c) -> str:
    """C expression (scalar type T, argument c) of a sympy expression built from + * pow exp log tanh sqrt and
    rational constants -- every node is checked; anything else raises (the kernel source is never user text)"""
    import sympy as sp

    if e == c:
        return "c"
    if e.is_Number:
        v = float(e)
        if not np.isfinite(v):
            raise UnsupportedClosureError(f"non-finite constant {e} in a closure")
        return f"T({v!r})"
    if isinstance(e, sp.Add):
        return "(" + " + ".join(_emit_c(a, c) for a in e.args) + ")"
    if isinstance(e, sp.Mul):
        return "(" + " * ".join(_emit_c(a, c) for a in e.args) + ")"
    if isinstance(e, sp.Pow):
        base, ex = e.args
        b = _emit_c(base, c)
        if ex.is_Integer:
            n = int(ex)
            if 1 <= n <= 16:
                return f"jit_powi({b}, {n})"
            if -16 <= n <= -1:
                return f"(T(1) / jit_powi({b}, {-n}))"
        if ex == sp.Rational(1, 2):
            return f"sqrt({b})"
        if ex == sp.Rational(-1, 2):
            return f"(T(1) / sqrt({b}))"
        return f"pow({b}, {_emit_c(ex, c)})"
    for fn, name in ((sp.exp, "exp"), (sp.log, "log"), (sp.tanh, "tanh")):
        if isinstance(e, fn):
            return f"{name}({_emit_c(e.args[0], c)})"
    raise UnsupportedClosureError(
        f"closure is outside the in-kernel family and uses {type(e).__name__}, which the run-time compiler does not take "
        "(+, *, powers, exp, log, tanh, sqrt, constants); non-pointwise closures such as CNNs are out of scope")


def jit_closure(expr, c, prelude: str = "") -> ClosureDesc:
    """``ClosureDesc(kind=JIT)`` of a traced pointwise expression: the body ``[prelude] return <expr>;`` for the run-time
    compiler and a numpy evaluator for host-side use (oracle comparisons, ``ClosureDesc.__call__``)"""
    import sympy as sp

    if expr.free_symbols - {c}:
        raise UnsupportedClosureError(f"closure depends on {expr.free_symbols - {c}}, not on the field alone")
    body = prelude + "return " + _emit_c(expr, c) + ";"
    if "\n" in body or len(body) > 16000:
        raise UnsupportedClosureError("closure expression too large for the run-time compiler")
    return ClosureDesc(JIT, 0, (0.0,), source=body, host_fn=sp.lambdify(c, expr, "numpy"))


def jit_body_of(desc: ClosureDesc) -> str:
    """C function body of ANY closure: a JIT closure's own, or the family member spelled out (when one role of a
    problem is compiled at run time the kernel is, and it has no table-driven evaluator: both roles get a body)"""
    if desc.kind == JIT:
        return desc.source
    co = [repr(float(v)) for v in desc.coef]
    if desc.kind == POLY:
        s = f"T r = T({co[-1]});" + "".join(f" r = r * c + T({a});" for a in co[-2::-1])
    else:  # the forward three-term recurrence of closures.hpp: series_generic
        s = f"const T x = T(2) * c - T(1); T r = T({co[0]});"
        if len(co) > 1:
            s += f" r += T({co[1]}) * x;"
        s += " T pm = T(1), pc = x;"
        for k in range(2, len(co)):
            s += f" {{ const T pn = (T({2 * k - 1}) * x * pc - T({k - 1}) * pm) / T({k}); r += T({co[k]}) * pn; pm = pc; pc = pn; }}"
    if desc.flags & LOGIT_PRIOR:
        s += " r += log(c / (T(1) - c));"
    if desc.flags & MIX_ENTROPY:
        s += " r += c * log(c) + (T(1) - c) * log(T(1) - c);"
    if desc.flags & EXP_WRAP:
        s += " r = exp(r);"
    return s + " return r;"


_trace_cache: dict = {}


def _cache_key(fn):
    """Hashable identity of a plain Python function: code object + captured scalars.  PDEEnv.step
    rebuilds the equation every step (pde_env.py:286) with the same callables (or the same lambda
    re-created around a new number); tracing through sympy costs ~1 ms, this lookup ~1 us."""
    code = getattr(fn, "__code__", None)
    if code is None:
        return None
    simple = (int, float, complex, str, bool, type(None))
    cells = []
    for c in getattr(fn, "__closure__", None) or ():
        try:
            v = c.cell_contents
        except ValueError:
            return None
        if not isinstance(v, simple):
            return None
        cells.append(v)
    defaults = getattr(fn, "__defaults__", None) or ()
    if not all(isinstance(v, simple) for v in defaults):
        return None
    if code.co_names and any(n not in ("np", "numpy", "jnp", "log", "exp", "ones_like", "zeros_like", "full_like",
                                       "sqrt", "square", "power") for n in code.co_names):
        return None  # refers to globals whose values may change between calls
    return (code, tuple(cells), tuple(defaults))


def as_closure(fn) -> ClosureDesc:
    """Descriptor for a user closure: ``ClosureDesc``, number, Legendre object or callable."""
    if isinstance(fn, ClosureDesc):
        return fn
    key = _cache_key(fn) if callable(fn) else None
    if key is not None:
        hit = _trace_cache.get(key)
        if hit is not None:
            return hit
        desc = _as_closure_uncached(fn)
        if len(_trace_cache) > 512:
            _trace_cache.clear()
        _trace_cache[key] = desc
        return desc
    return _as_closure_uncached(fn)


def _as_closure_uncached(fn) -> ClosureDesc:
    if isinstance(fn, ClosureDesc):
        return fn
    if hasattr(fn, "closure_desc"):
        return fn.closure_desc()
    if isinstance(fn, numbers.Real):
        return constant(float(fn))
    if not callable(fn):
        raise UnsupportedClosureError(f"cannot interpret {fn!r} as a pointwise closure")
    import sympy as sp

    c = sp.Symbol("c", real=True)
    try:
        out = fn(_Sym(c))
    except UnsupportedClosureError:
        raise
    except Exception as e:  # the callable did something the tracer cannot follow
        raise UnsupportedClosureError(
            f"could not trace closure {fn!r} symbolically ({type(e).__name__}: {e}); "
            "use operators / numpy ufuncs (np.log, np.exp) on the argument, or pass a ClosureDesc"
        ) from e
    if isinstance(out, _Sym):
        return _match(out.e, c)
    if isinstance(out, numbers.Real):
        return constant(float(out))
    if isinstance(out, np.ndarray) and out.ndim == 0:
        return constant(float(out))
    raise UnsupportedClosureError(f"closure returned {type(out).__name__}, expected an array expression")


def poly_in_t(fn, samples=(0.0, 0.37, 1.9), max_degree: int = 3):
    """Ascending coefficients of ``fn(t)`` if it is a polynomial in t of degree <= ``max_degree`` (a number, or a
    callable built from operators on t -- the ``theta(t)`` / ``flux(t)`` fields of the smoothed-boundary equations,
    cahn_hilliard.py:232-235; notebooks/smooth_boundary.ipynb:262 is a quadratic), else ``None``.  The traced
    polynomial is checked against the callable itself at a few times before it is trusted: the in-kernel adaptive
    solve evaluates it at stage times of its own choosing (``pdeopt_set_time_terms_poly``)."""
    if isinstance(fn, numbers.Real):
        return [float(fn)]
    if not callable(fn):
        return None
    import sympy as sp

    t = sp.Symbol("t", real=True)
    try:
        out = fn(_Sym(t))
    except Exception:
        return None
    if isinstance(out, _Sym):
        try:
            poly = sp.Poly(sp.expand(out.e), t)
        except Exception:
            return None
        if poly.degree() > max_degree or any(not c.is_number for c in poly.all_coeffs()):
            return None
        coef = [float(c) for c in reversed(poly.all_coeffs())]
    elif isinstance(out, numbers.Real) or (isinstance(out, np.ndarray) and out.ndim == 0):
        coef = [float(out)]
    else:
        return None
    if not all(np.isfinite(coef)):
        return None
    for ts in samples:  # the callable has the last word
        try:
            want = float(fn(float(ts)))
        except Exception:
            return None
        got = sum(c * ts**i for i, c in enumerate(coef))
        if not abs(got - want) <= 1e-12 * max(1.0, abs(want)):
            return None
    return coef
