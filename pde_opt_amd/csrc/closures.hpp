// Pointwise closure family evaluated inside the stencil kernels (device code).
//
// Stands in for the reference's Python callables mu(u), D(u), R(u) (dataclass fields
// pde_opt/numerics/equations/cahn_hilliard.py:51-54, allen_cahn.py:47-50; catalogue SURVEY
// Appendix D).  See include/pdeopt_hip.h for the definition of the family.
#pragma once

#include "common.hpp"

namespace pdeopt {

template <typename T>
__device__ __forceinline__ T t_log(T x);
// fp32: the hardware log2 (v_log_f32, 1 ulp) instead of ocml's logf (~40 VALU instructions with its range fix-ups), as
// for t_logit below: the only user is the mixing-entropy term c ln c + (1 - c) ln(1 - c) of the smoothed-boundary free
// energy, whose ~1e-7 absolute error is below the fp32 noise of the stencil it feeds.  fp64 stays libm's.
template <>
__device__ __forceinline__ float t_log<float>(float x) {
  return 0.6931471805599453f * __builtin_amdgcn_logf(x);
}
template <>
__device__ __forceinline__ double t_log<double>(double x) {
  return log(x);
}
template <typename T>
__device__ __forceinline__ T t_exp(T x);
template <>
__device__ __forceinline__ float t_exp<float>(float x) {
  return expf(x);
}
template <>
__device__ __forceinline__ double t_exp<double>(double x) {
  return exp(x);
}

// 1 / x and sqrt(x) of the smoothed-boundary kernels (kappa / psi, ... / psi, sqrt(2 f)): fp32 on the hardware
// approximations (v_rcp_f32 / v_sqrt_f32, 1 ulp; the IEEE division and square root expand to ~10 instructions each with
// their scaling and fix-up sequences), fp64 exact
template <typename T>
__device__ __forceinline__ T t_rcp(T x);
template <>
__device__ __forceinline__ float t_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <>
__device__ __forceinline__ double t_rcp<double>(double x) { return 1.0 / x; }
template <typename T>
__device__ __forceinline__ T t_sqrt(T x);
template <>
__device__ __forceinline__ float t_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <>
__device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }

// log(c / (1 - c)).  fp32: hardware reciprocal and log2 (v_rcp_f32 / v_log_f32, 1 ulp each) instead of
// the correctly rounded division + ocml logf (~40 VALU instructions with their range fix-ups): the
// fused kernels are VALU-bound, and the added error (~1e-7 absolute on a value of O(1)) is below the
// fp32 noise floor of the stencil it feeds.  fp64 stays exact.
template <typename T>
__device__ __forceinline__ T t_logit(T c);
template <>
__device__ __forceinline__ float t_logit<float>(float c) {
  return 0.6931471805599453f * __builtin_amdgcn_logf(c * __builtin_amdgcn_rcpf(1.0f - c));
}
// fp64: log r, r = c / (1 - c) (correctly rounded division), by range reduction with the hardware frexp and the
// atanh series  log m = 2 s (1 + z/3 + z^2/5 + ...),  s = (m - 1) / (m + 1),  z = s^2,  m in [sqrt(1/2), sqrt 2):
// |s| <= 0.172, nine terms reach 4e-16 relative / 2e-15 absolute error over c in (1e-6, 1 - 1e-6) -- the accuracy
// class of ocml's log (checked against 40-digit arithmetic, tools/logit64_check.py) in ~45 instead of ~90
// instructions: ocml's double-double log was 60 % of the fp64 pair kernel's VALU instructions.  Values outside
// (0, 1) keep log's conventions (c = 0: -inf, c = 1: +inf, otherwise NaN) so a diverged run still reports NaNs.
#ifndef PDEOPT_LOGIT64_OCML
#ifndef PDEOPT_LOGIT64_TWO_DIV
// ONE division: with c = ma 2^ea and 1 - c = mb 2^eb (hardware frexp, mantissas in [0.5, 1)) the ratio is
// (ma / mb) 2^(ea - eb); ma is doubled / halved so that m = ma' / mb lands in [0.707, 1.414) without ever forming it,
// and s = (m - 1) / (m + 1) = (ma' - mb) / (ma' + mb) directly -- the numerator is exact (Sterbenz), so s carries one
// rounding of the sum and one of the quotient.  (The first form divided twice, c / (1 - c) and (m - 1) / (m + 1):
// a v_rcp_f64 is 17 clocks, profiles/r02_valubench_raw.txt; -DPDEOPT_LOGIT64_TWO_DIV keeps it.)
template <>
__device__ __forceinline__ double t_logit<double>(double c) {
  const double b = 1.0 - c;
  int ea, eb;
  double ma = frexp(c, &ea);
  const double mb = frexp(b, &eb);
  const bool lo = ma < 0.70710678118654752440 * mb;  // ratio below sqrt(1/2): double it
  const bool hi = ma > 1.41421356237309504880 * mb;  // above sqrt(2): halve it
  const int k = lo ? 1 : (hi ? -1 : 0);              // one 32-bit select pair instead of four 64-bit halves
  ma = ldexp(ma, k);
  const int e = ea - eb - k;
  // (ma - mb) / (ma + mb) with the denominator in [1, 3): no scaling, no special cases -- reciprocal, two Newton
  // steps, quotient, one residual correction (8 instructions against the 11-12 of the IEEE sequence with its
  // v_div_scale / v_div_fmas / v_div_fixup; the result is within an ulp either way)
  const double num = ma - mb, den = ma + mb;
#ifdef PDEOPT_LOGIT64_IEEE_DIV
  const double s = num / den;
#else
  double rc = __builtin_amdgcn_rcp(den);
  rc = __builtin_fma(rc, __builtin_fma(-den, rc, 1.0), rc);
  rc = __builtin_fma(rc, __builtin_fma(-den, rc, 1.0), rc);
  const double q0 = num * rc;
  const double s = __builtin_fma(__builtin_fma(-den, q0, num), rc, q0);
#endif
  const double z = s * s;
  double p = 1.0 / 19.0;
  p = p * z + 1.0 / 17.0;
  p = p * z + 1.0 / 15.0;
  p = p * z + 1.0 / 13.0;
  p = p * z + 1.0 / 11.0;
  p = p * z + 1.0 / 9.0;
  p = p * z + 1.0 / 7.0;
  p = p * z + 1.0 / 5.0;
  p = p * z + 1.0 / 3.0;
  const double two_s = s + s;
  const double lm = two_s + two_s * (p * z);
  const double ed = (double)e;
  double res = ed * 6.93147180369123816490e-01 + (lm + ed * 1.90821492927058770002e-10);
  if (!(c > 0.0 && c < 1.0)) res = (c == 0.0) ? -INFINITY : ((c == 1.0) ? INFINITY : NAN);
  return res;
}
#else
template <>
__device__ __forceinline__ double t_logit<double>(double c) {
  const double r = c / (1.0 - c);
  int e;
  double m = frexp(r, &e);  // m in [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? 2.0 * m : m;
  e = lo ? e - 1 : e;
  const double s = (m - 1.0) / (m + 1.0);
  const double z = s * s;
  double p = 1.0 / 19.0;
  p = p * z + 1.0 / 17.0;
  p = p * z + 1.0 / 15.0;
  p = p * z + 1.0 / 13.0;
  p = p * z + 1.0 / 11.0;
  p = p * z + 1.0 / 9.0;
  p = p * z + 1.0 / 7.0;
  p = p * z + 1.0 / 5.0;
  p = p * z + 1.0 / 3.0;
  const double two_s = s + s;
  const double lm = two_s + two_s * (p * z);
  const double ed = (double)e;
  double res = ed * 6.93147180369123816490e-01 + (lm + ed * 1.90821492927058770002e-10);
  if (!(r > 0.0) || r == INFINITY) res = (r == 0.0) ? -INFINITY : ((r == INFINITY) ? INFINITY : NAN);
  return res;
}
#endif
#else
template <>
__device__ __forceinline__ double t_logit<double>(double c) {
  return log(c / (1.0 - c));
}
#endif

// Closure specialisation classes (template parameter CL of the kernels):
//   CL_GENERIC : any kind / flags / n, decided at run time (wave-uniform branches)
//   CL_POLY    : mu = cubic polynomial, mobility = quadratic polynomial, fully unrolled
//   CL_LOGIT   : as CL_POLY with the log(c/(1-c)) prior added to mu (regular-solution model)
//   CL_LOGIT1  : CL_LOGIT whose polynomial part is linear (the regular-solution model 3 (1 - 2c) of the
//                headline workload): two FMAs less per evaluation in the VALU-bound fused CH kernel
//                (coefficients past n are stored as zeros), and the fused fp32/fp64 pair kernel folds
//                -kappa lap into the same expression (stencil_fused.hpp, FOLD_MU)
//   CL_POLY_M0 : CL_POLY with a constant mobility (Allen-Cahn's R = 1, BASELINE config 2): two FMAs less
//                per evaluation in the VALU-bound single-pass RK4 kernel, bitwise equal on finite states
enum { CL_GENERIC = 0, CL_POLY = 1, CL_LOGIT = 2, CL_LOGIT1 = 3, CL_POLY_M0 = 4 };

template <typename T>
__device__ __forceinline__ T series_generic(const ClosureSpec& s, const T* __restrict__ coef, T c) {
  T r;
  if (s.kind == PDEOPT_CL_POLY) {
    r = coef[s.n - 1];
    for (int k = s.n - 2; k >= 0; --k) r = r * c + coef[k];
  } else {
    // forward three-term recurrence on x = 2c - 1, the order legendre.py:23-34 uses
    const T x = T(2) * c - T(1);
    r = coef[0];
    if (s.n > 1) r += coef[1] * x;
    T pm = T(1), pc = x;
    for (int k = 2; k < s.n; ++k) {
      const T pn = (T(2 * k - 1) * x * pc - T(k - 1) * pm) / T(k);
      r += coef[k] * pn;
      pm = pc;
      pc = pn;
    }
  }
  return r;
}

template <typename T>
__device__ __forceinline__ T closure_generic(const ClosureSpec& s, const T* __restrict__ coef, T c) {
  T r = series_generic<T>(s, coef, c);
  if (s.flags & PDEOPT_CL_LOGIT_PRIOR) r += t_logit<T>(c);
  if (s.flags & PDEOPT_CL_MIX_ENTROPY) r += c * t_log<T>(c) + (T(1) - c) * t_log<T>(T(1) - c);
  if (s.flags & PDEOPT_CL_EXP_WRAP) r = t_exp<T>(r);
  return r;
}

// mu_h(c)
template <typename T, int CL>
__device__ __forceinline__ T eval_mu(const ClosureSpec& s, const T* __restrict__ coef, T c) {
  if constexpr (CL == CL_GENERIC) {
    return closure_generic<T>(s, coef, c);
  } else if constexpr (CL == CL_LOGIT1) {
    return coef[1] * c + coef[0] + t_logit<T>(c);
  } else {
    T r = ((coef[3] * c + coef[2]) * c + coef[1]) * c + coef[0];
    if constexpr (CL == CL_LOGIT) r += t_logit<T>(c);
    return r;
  }
}

// D(c) or R(c)
template <typename T, int CL>
__device__ __forceinline__ T eval_mob(const ClosureSpec& s, const T* __restrict__ coef, T c) {
  if constexpr (CL == CL_GENERIC) {
    return closure_generic<T>(s, coef, c);
  } else if constexpr (CL == CL_POLY_M0) {
    return coef[0];
  } else {
    return (coef[2] * c + coef[1]) * c + coef[0];
  }
}

// host side: which specialisation covers a (mu, mob) pair
inline int classify_closures(const pdeopt_closure& mu, const pdeopt_closure& mob) {
  const bool mob_ok = mob.kind == PDEOPT_CL_POLY && mob.flags == 0 && mob.n <= 3;
  const bool mu_poly = mu.kind == PDEOPT_CL_POLY && mu.n <= 4 &&
                       (mu.flags & ~PDEOPT_CL_LOGIT_PRIOR) == 0;
  if (mob_ok && mu_poly) return (mu.flags & PDEOPT_CL_LOGIT_PRIOR) ? CL_LOGIT : CL_POLY;
  return CL_GENERIC;
}

}  // namespace pdeopt
