/*
 * c_oracle.c -- plain-C restatement of the pde_opt hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Second, independent checker next to oracle/np_oracle.py (which mirrors the reference's roll form
 * op for op): here the same index-form arithmetic is written as fused
 * single-pass loops, the way an optimised CPU implementation would do it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (pde_opt_amd/) never does.
 *
 * Reference lines followed:
 *   lap5            pde_opt/numerics/utils/derivatives.py:8-12
 *   mu              pde_opt/numerics/equations/cahn_hilliard.py:93, allen_cahn.py:83
 *   face grad/avg   derivatives.py:24-31, 39-46      flux: cahn_hilliard.py:105-106
 *   divergence      derivatives.py:54-61, cahn_hilliard.py:109
 *   Allen-Cahn      allen_cahn.py:81-84
 *   RK4             not in the reference (textbook tableau; parity pinned through np_oracle)
 * Closures: the same family as include/pdeopt_hip.h (poly / Legendre, optional logit prior,
 * optional exp), evaluated in the arithmetic type T.
 *
 * build: make -C oracle      (gcc -O2 -fopenmp; -ffp-contract=off keeps it free of FMA re-rounding)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int32_t kind, flags, n, reserved;
  double coef[16];
} oracle_closure;

#define CL_POLY 0
#define CL_LEGENDRE 1
#define CL_LOGIT 1
#define CL_EXP 2

#define DEFINE_ORACLE(T, SUF, LOGF, EXPF)                                                          \
  static inline T closure_##SUF(const oracle_closure* c, T x) {                                    \
    T r;                                                                                           \
    if (c->kind == CL_POLY) {                                                                      \
      r = (T)c->coef[c->n - 1];                                                                    \
      for (int k = c->n - 2; k >= 0; --k) r = r * x + (T)c->coef[k];                               \
    } else {                                                                                       \
      const T z = (T)2 * x - (T)1;                                                                 \
      r = (T)c->coef[0];                                                                           \
      if (c->n > 1) r += (T)c->coef[1] * z;                                                        \
      T pm = (T)1, pc = z;                                                                         \
      for (int k = 2; k < c->n; ++k) {                                                             \
        const T pn = ((T)(2 * k - 1) * z * pc - (T)(k - 1) * pm) / (T)k;                           \
        r += (T)c->coef[k] * pn;                                                                   \
        pm = pc;                                                                                   \
        pc = pn;                                                                                   \
      }                                                                                            \
    }                                                                                              \
    if (c->flags & CL_LOGIT) r += LOGF(x / ((T)1 - x));                                            \
    if (c->flags & CL_EXP) r = EXPF(r);                                                            \
    return r;                                                                                      \
  }                                                                                                \
                                                                                                   \
  /* mu[i,j] = mu_h(u) - kappa lap5(u) on the whole grid (periodic) */                             \
  static void chem_potential_##SUF(const T* u, T* mu, int nx, int ny, T hx, T hy, T kappa,         \
                                   const oracle_closure* cmu) {                                    \
    _Pragma("omp parallel for schedule(static)") for (int i = 0; i < nx; ++i) {                    \
      const int ip = (i + 1) % nx, im = (i + nx - 1) % nx;                                         \
      for (int j = 0; j < ny; ++j) {                                                               \
        const int jp = (j + 1) % ny, jm = (j + ny - 1) % ny;                                       \
        const T c = u[(size_t)i * ny + j];                                                         \
        const T lap = (u[(size_t)ip * ny + j] - 2 * c + u[(size_t)im * ny + j]) / (hx * hx) +      \
                      (u[(size_t)i * ny + jp] - 2 * c + u[(size_t)i * ny + jm]) / (hy * hy);       \
        mu[(size_t)i * ny + j] = closure_##SUF(cmu, c) - kappa * lap;                              \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
                                                                                                   \
  /* out = rhs(u);  eq 0 = Cahn-Hilliard, 1 = Allen-Cahn.  work: nx*ny scratch (CH: mu, then D) */ \
  void oracle_rhs_##SUF(int eq, const T* u, T* out, T* work, T* work2, int nx, int ny, double hx_, \
                        double hy_, double kappa_, const oracle_closure* cmu,                      \
                        const oracle_closure* cmob) {                                              \
    const T hx = (T)hx_, hy = (T)hy_, kappa = (T)kappa_;                                           \
    chem_potential_##SUF(u, work, nx, ny, hx, hy, kappa, cmu);                                     \
    if (eq == 1) {                                                                                 \
      _Pragma("omp parallel for schedule(static)") for (size_t p = 0; p < (size_t)nx * ny; ++p)    \
          out[p] = -closure_##SUF(cmob, u[p]) * work[p];                                           \
      return;                                                                                      \
    }                                                                                              \
    _Pragma("omp parallel for schedule(static)") for (size_t p = 0; p < (size_t)nx * ny; ++p)      \
        work2[p] = closure_##SUF(cmob, u[p]);                                                      \
    const T* mu = work;                                                                            \
    const T* D = work2;                                                                            \
    _Pragma("omp parallel for schedule(static)") for (int i = 0; i < nx; ++i) {                    \
      const int ip = (i + 1) % nx, im = (i + nx - 1) % nx;                                         \
      for (int j = 0; j < ny; ++j) {                                                               \
        const int jp = (j + 1) % ny, jm = (j + ny - 1) % ny;                                       \
        const size_t c = (size_t)i * ny + j;                                                       \
        const size_t xp = (size_t)ip * ny + j, xm = (size_t)im * ny + j;                           \
        const size_t yp = (size_t)i * ny + jp, ym = (size_t)i * ny + jm;                           \
        const T fx0 = ((T)0.5 * (D[c] + D[xp])) * ((mu[xp] - mu[c]) / hx);                         \
        const T fxm = ((T)0.5 * (D[xm] + D[c])) * ((mu[c] - mu[xm]) / hx);                         \
        const T fy0 = ((T)0.5 * (D[c] + D[yp])) * ((mu[yp] - mu[c]) / hy);                         \
        const T fym = ((T)0.5 * (D[ym] + D[c])) * ((mu[c] - mu[ym]) / hy);                         \
        out[c] = (fx0 - fxm) / hx + (fy0 - fym) / hy;                                              \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
                                                                                                   \
  /* n classical RK4 substeps in place; scratch = 5 * nx*ny elements */                            \
  void oracle_rk4_##SUF(int eq, T* y, T* scratch, int nx, int ny, double hx, double hy,            \
                        double kappa, const oracle_closure* cmu, const oracle_closure* cmob,       \
                        double dt_, int64_t n) {                                                   \
    const size_t N = (size_t)nx * ny;                                                              \
    T *k = scratch, *ys = scratch + N, *acc = scratch + 2 * N, *w1 = scratch + 3 * N,              \
      *w2 = scratch + 4 * N;                                                                       \
    const T dt = (T)dt_;                                                                           \
    for (int64_t s = 0; s < n; ++s) {                                                              \
      oracle_rhs_##SUF(eq, y, k, w1, w2, nx, ny, hx, hy, kappa, cmu, cmob);                        \
      _Pragma("omp parallel for schedule(static)") for (size_t p = 0; p < N; ++p) {                \
        acc[p] = y[p] + (dt / 6) * k[p];                                                           \
        ys[p] = y[p] + (dt / 2) * k[p];                                                            \
      }                                                                                            \
      oracle_rhs_##SUF(eq, ys, k, w1, w2, nx, ny, hx, hy, kappa, cmu, cmob);                       \
      _Pragma("omp parallel for schedule(static)") for (size_t p = 0; p < N; ++p) {                \
        acc[p] = acc[p] + (dt / 3) * k[p];                                                         \
        ys[p] = y[p] + (dt / 2) * k[p];                                                            \
      }                                                                                            \
      oracle_rhs_##SUF(eq, ys, k, w1, w2, nx, ny, hx, hy, kappa, cmu, cmob);                       \
      _Pragma("omp parallel for schedule(static)") for (size_t p = 0; p < N; ++p) {                \
        acc[p] = acc[p] + (dt / 3) * k[p];                                                         \
        ys[p] = y[p] + dt * k[p];                                                                  \
      }                                                                                            \
      oracle_rhs_##SUF(eq, ys, k, w1, w2, nx, ny, hx, hy, kappa, cmu, cmob);                       \
      _Pragma("omp parallel for schedule(static)") for (size_t p = 0; p < N; ++p)                  \
          y[p] = acc[p] + (dt / 6) * k[p];                                                         \
    }                                                                                              \
  }

DEFINE_ORACLE(float, f32, logf, expf)
DEFINE_ORACLE(double, f64, log, exp)

#ifdef _OPENMP
#include <omp.h>
int oracle_max_threads(void) { return omp_get_max_threads(); }
void oracle_set_threads(int n) { omp_set_num_threads(n); }
#else
int oracle_max_threads(void) { return 1; }
void oracle_set_threads(int n) { (void)n; }
#endif
