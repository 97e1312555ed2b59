// Generic (untiled) fused RHS + integrator-stage kernel: any grid shape (including Nx or Ny of
// 1, 2, 3 where periodic neighbours coincide), any closure, fp32 and fp64.  One thread per cell,
// neighbours fetched straight from global memory (L1/L2 supply the reuse).  This is the
// correctness anchor and the fallback for shapes the LDS-tiled kernels do not cover; it is a
// HIP kernel, not a CPU path.
//
// Arithmetic follows, in index form:
//   lap, mu              pde_opt/numerics/utils/derivatives.py:8-12, cahn_hilliard.py:93
//   face grad / average  derivatives.py:24-31, 39-46
//   divergence           derivatives.py:54-61, cahn_hilliard.py:105-109
//   Allen-Cahn           allen_cahn.py:81-84
#pragma once

#include "closures.hpp"

namespace pdeopt {

// out/acc update performed after k = rhs(in) has been formed for a cell
// OUT_K_LC: out = k, and next = y + sum_j c_j K_j + c_n k  (the input of the following stage of an
// embedded Runge-Kutta method, formed where k is still in registers; Tsit5)
enum { OUT_NONE = 0, OUT_K = 1, OUT_Y_PLUS_AK = 2, OUT_ACC_PLUS_BK = 3, OUT_K_LC = 4 };
enum { ACC_NONE = 0, ACC_INIT = 1, ACC_ADD = 2 };

constexpr int kMaxLc = 6;
template <typename T>
struct LcArgs {
  const T* k[kMaxLc];  // earlier slopes
  T c[kMaxLc + 1];     // their coefficients (already multiplied by dt); c[n] belongs to this stage's k
  int n;
  T* next;
};

template <typename T>
struct StageArgs {
  const T* in;   // field the RHS is evaluated on (stencil reads)
  const T* y;    // base state of the substep (pointwise)
  T* out;        // primary output (pointwise)
  T* acc;        // running RK accumulator (pointwise)
  T a, b;        // already multiplied by dt
  T rhx, rhy, rhx2, rhy2;
  T rhz, rhz2;   // 3-D equations
  const T* mu3;  // 3-D: chemical potential field of the first pass
  Geo g;
  const EnvParams<T>* ep;
  ClosureSpec mu, mob;
  const T* vx;   // advection-diffusion face velocities
  const T* vy;
  int64_t vstride;
  const T* psi;  // smoothed-boundary level set, |grad psi|/psi and wall mask (shared [nx][ny])
  const T* ngp;
  const T* mask;
  T tw_a, tw_b, tsrc;  // cos(theta(t)) on / off the mask, flux(t): scalars of this RHS evaluation
  ClosureSpec fe;
  LcArgs<T> lc;  // OUT_K_LC only
  int out_mode, acc_mode;
  int scaled;  // multiply k by EnvParams::kscale (per-environment step sizes of the adaptive driver)
  int dbg;  // timing ablations (PDEOPT_OPT_DEBUG_ABLATE): bit0 skip mu phase, bit1 skip flux phase
};

// Workgroup -> (tile row, tile column, environment).  `flags` (host: tile_flags()): bit 0 = XCD-aware block map,
// bits 8-15 / 16-23 = log2(tiles_j) + 1 / log2(tiles_i) + 1 when both are powers of two.  The shift form matters:
// the quotients are wave-uniform, but gfx950 has no scalar integer division, so `t % tiles_j` etc. expand into
// ~20 VALU instructions each (v_cvt / v_rcp_iflag / v_mul_hi ...) -- 3 divisions were 8 % of the stage-pair
// kernel's VALU instructions.
__device__ __forceinline__ void decode_tile(int t, int tiles_i, int tiles_j, int nblk, int flags, int* ti, int* tj, int* b) {
  if (flags & 1) t = (t & 7) * (nblk >> 3) + (t >> 3);
#ifdef PDEOPT_TILE_DIV  // A/B build: always the division form
  const int sj = 0, si = 0;
#else
  const int sj = (flags >> 8) & 0xff, si = (flags >> 16) & 0xff;
#endif
  if (sj && si) {
    *tj = t & (tiles_j - 1);
    const int q = t >> (sj - 1);
    *ti = q & (tiles_i - 1);
    *b = q >> (si - 1);
  } else {
    *tj = t % tiles_j;
    *ti = (t / tiles_j) % tiles_i;
    *b = t / (tiles_j * tiles_i);
  }
  // the division form computes in VGPRs; tell the compiler the results are wave-uniform either way, so that
  // everything derived from them (environment base offsets, tile origins, row wraps) is scalar work
  *tj = __builtin_amdgcn_readfirstlane(*tj);
  *ti = __builtin_amdgcn_readfirstlane(*ti);
  *b = __builtin_amdgcn_readfirstlane(*b);
}
inline int tile_flags(int nblk, int tiles_i, int tiles_j) {
  auto lg = [](int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
  };
  int f = (nblk % 8 == 0) ? 1 : 0;
  const int li = lg(tiles_i), lj = lg(tiles_j);
  if (li >= 0 && lj >= 0) f |= ((lj + 1) << 8) | ((li + 1) << 16);
  return f;
}

// Tile + halo -> LDS, one tile row per wave and trip: lane l < PV loads vector l (V elements) of a row.  The row
// index is wave-uniform, so its periodic wrap and the row offset are scalar work and a trip costs the VALU nothing
// but the load and the LDS store; the column wrap and offset are formed once per thread.  (A flat "vector idx =
// tid + it NT" mapping needs a division by PV, two wraps and a 64-bit address per vector: it was ~14 % of the
// VALU instructions of the VALU-bound fused kernels.)  All loads are issued before the stores; a wave past the last
// row re-reads that row and skips the store.  sdst: LDS row 0 / vector 0; P: LDS row pitch in elements;
// i_first / j_first: global row / column of LDS row 0 / vector 0.
template <typename T, int V, int PV, int NT, int ROWS, typename Vec, typename WR, typename WC>
__device__ __forceinline__ void load_rows_per_wave(T* __restrict__ sdst, const int P, const T* __restrict__ in, const int64_t ld,
                                                   const int i_first, const int j_first, WR wrap_row, WC wrap_col,
                                                   const int tid) {
  constexpr int NW = NT / 64, kTrips = (ROWS + NW - 1) / NW;
  static_assert(PV <= 64, "a row of vectors fits one wave");
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  if (lane < PV) {
    const T* __restrict__ colp = in + wrap_col(j_first + lane * V);
    T* const lds = sdst + lane * V;
    Vec f[kTrips];
#pragma unroll
    for (int k = 0; k < kTrips; ++k) {
      int row = wave + k * NW;
      if constexpr (ROWS % NW != 0) row = row < ROWS ? row : ROWS - 1;
      f[k] = *reinterpret_cast<const Vec*>(colp + (int64_t)wrap_row(i_first + row) * ld);
    }
#pragma unroll
    for (int k = 0; k < kTrips; ++k) {
      const int row = wave + k * NW;
      if (ROWS % NW == 0 || k + 1 < kTrips || row < ROWS) *reinterpret_cast<Vec*>(lds + row * P) = f[k];
    }
  }
}

__device__ __forceinline__ int wrap_idx(int i, int n) {
  i %= n;
  return i < 0 ? i + n : i;
}

template <typename T>
__device__ __forceinline__ T lap_at(T c, T xp, T xm, T yp, T ym, T rhx2, T rhy2) {
  return (xp - T(2) * c + xm) * rhx2 + (yp - T(2) * c + ym) * rhy2;
}

template <typename T, int EQ>
__device__ __forceinline__ T rhs_generic_point(const StageArgs<T>& a, const T* __restrict__ u,
                                               const EnvParams<T>& p, int i, int j, int b) {
  const Geo& g = a.g;
  int i1 = i + 1, i2 = i + 2, im1 = i - 1, im2 = i - 2;
  int j1 = j + 1, j2 = j + 2, jm1 = j - 1, jm2 = j - 2;
  if (g.periodic) {
    i1 = wrap_idx(i1, g.nx); i2 = wrap_idx(i2, g.nx);
    im1 = wrap_idx(im1, g.nx); im2 = wrap_idx(im2, g.nx);
    j1 = wrap_idx(j1, g.ny); j2 = wrap_idx(j2, g.ny);
    jm1 = wrap_idx(jm1, g.ny); jm2 = wrap_idx(jm2, g.ny);
  }
  const int64_t ld = g.ld;
  auto U = [&](int ii, int jj) -> T { return u[(int64_t)ii * ld + jj]; };
  const T u00 = U(i, j), uxp = U(i1, j), uxm = U(im1, j), uyp = U(i, j1), uym = U(i, jm1);

  if constexpr (EQ == PDEOPT_EQ_ALLEN_CAHN) {
    const T mu = closure_generic<T>(a.mu, p.mu, u00) -
                 p.kappa * lap_at<T>(u00, uxp, uxm, uyp, uym, a.rhx2, a.rhy2);
    return -closure_generic<T>(a.mob, p.mob, u00) * mu;
  } else if constexpr (EQ == PDEOPT_EQ_ALLEN_CAHN_SBM || EQ == PDEOPT_EQ_CAHN_HILLIARD_SBM) {
    // smoothed-boundary method (allen_cahn.py:139-156, cahn_hilliard.py:257-289): psi-weighted
    // Laplacian through face averages, wall (contact-angle) term, optional boundary flux source
    const int ny = g.ny;
    auto PS = [&](int ii, int jj) -> T { return a.psi[(int64_t)ii * ny + jj]; };
    const T sqk = sqrt(p.kappa);
    auto WL = [&](int ii, int jj) -> T {
      const T m = a.mask[(int64_t)ii * ny + jj];
      return sqk * a.ngp[(int64_t)ii * ny + jj] * (a.tw_a * m + a.tw_b * (T(1) - m));
    };
    // inner(c; neighbours) = mu_h(c) - kappa/psi div(psi grad u) - wall sqrt(2 f(c))
    auto INNER = [&](T c, T xp, T xm, T yp, T ym, T pc, T pxp, T pxm, T pyp, T pym, T w) -> T {
      const T dx_hi = (T(0.5) * (pc + pxp)) * ((xp - c) * a.rhx), dx_lo = (T(0.5) * (pxm + pc)) * ((c - xm) * a.rhx);
      const T dy_hi = (T(0.5) * (pc + pyp)) * ((yp - c) * a.rhy), dy_lo = (T(0.5) * (pym + pc)) * ((c - ym) * a.rhy);
      const T lap = (dx_hi - dx_lo) * a.rhx + (dy_hi - dy_lo) * a.rhy;
      T r = closure_generic<T>(a.mu, p.mu, c) - (p.kappa / pc) * lap;
      r -= w * sqrt(T(2) * closure_generic<T>(a.fe, p.fe, c));
      return r;
    };
    const T p00 = PS(i, j), pxp = PS(i1, j), pxm = PS(im1, j), pyp = PS(i, j1), pym = PS(i, jm1);
    const T in00 = INNER(u00, uxp, uxm, uyp, uym, p00, pxp, pxm, pyp, pym, WL(i, j));
    if constexpr (EQ == PDEOPT_EQ_ALLEN_CAHN_SBM) {
      return -closure_generic<T>(a.mob, p.mob, u00) * in00;
    } else {
      const T ux2 = U(i2, j), uxm2 = U(im2, j), uy2 = U(i, j2), uym2 = U(i, jm2);
      const T upp = U(i1, j1), upm = U(i1, jm1), ump = U(im1, j1), umm = U(im1, jm1);
      const T px2 = PS(i2, j), pxm2 = PS(im2, j), py2 = PS(i, j2), pym2 = PS(i, jm2);
      const T ppp = PS(i1, j1), ppm = PS(i1, jm1), pmp = PS(im1, j1), pmm = PS(im1, jm1);
      const T inxp = INNER(uxp, ux2, u00, upp, upm, pxp, px2, p00, ppp, ppm, WL(i1, j));
      const T inxm = INNER(uxm, u00, uxm2, ump, umm, pxm, p00, pxm2, pmp, pmm, WL(im1, j));
      const T inyp = INNER(uyp, upp, ump, uy2, u00, pyp, ppp, pmp, py2, p00, WL(i, j1));
      const T inym = INNER(uym, upm, umm, u00, uym2, pym, ppm, pmm, p00, pym2, WL(i, jm1));
      const T d00 = closure_generic<T>(a.mob, p.mob, u00);
      const T dxp = closure_generic<T>(a.mob, p.mob, uxp), dxm = closure_generic<T>(a.mob, p.mob, uxm);
      const T dyp = closure_generic<T>(a.mob, p.mob, uyp), dym = closure_generic<T>(a.mob, p.mob, uym);
      const T fx0 = (T(0.5) * (p00 + pxp)) * (T(0.5) * (d00 + dxp)) * ((inxp - in00) * a.rhx);
      const T fxm = (T(0.5) * (pxm + p00)) * (T(0.5) * (dxm + d00)) * ((in00 - inxm) * a.rhx);
      const T fy0 = (T(0.5) * (p00 + pyp)) * (T(0.5) * (d00 + dyp)) * ((inyp - in00) * a.rhy);
      const T fym = (T(0.5) * (pym + p00)) * (T(0.5) * (dym + d00)) * ((in00 - inym) * a.rhy);
      return ((fx0 - fxm) * a.rhx + (fy0 - fym) * a.rhy) / p00 + a.ngp[(int64_t)i * ny + j] * a.tsrc;
    }
  } else if constexpr (EQ == PDEOPT_EQ_SHAPE_SMOOTH) {
    // Shape.smooth_shape (shapes.py:44-64): centred first / second / mixed differences (derivatives.py:62-106),
    // u_nn = (u_xx u_x^2 + 2 u_xy u_x u_y + u_yy u_y^2) / |grad u|^2 with |grad u|^2 < 1e-7 replaced by 1
    const T upp = U(i1, j1), upm = U(i1, jm1), ump = U(im1, j1), umm = U(im1, jm1);
    const T gx = T(0.5) * (uxp - uxm) * a.rhx, gy = T(0.5) * (uyp - uym) * a.rhy;
    const T gxx = (uxp - T(2) * u00 + uxm) * a.rhx2, gyy = (uyp - T(2) * u00 + uym) * a.rhy2;
    const T gxy = (upp + umm - ump - upm) * (T(0.25) * a.rhx * a.rhy);
    T g2 = gx * gx + gy * gy;
    if (g2 < T(1e-7)) g2 = T(1);
    const T unn = (gxx * gx * gx + T(2) * gxy * gx * gy + gyy * gy * gy) / g2;
    const T c = p.kappa, eps = p.gpe_k;
    const T pot = T(18) / eps * u00 * (T(1) - u00) * (T(1) - T(2) * u00);
    return T(2) * (c * (gxx + gyy) + (T(1) - c) * unn) - pot / eps;
  } else if constexpr (EQ == PDEOPT_EQ_ADVECTION_DIFFUSION) {
    const int64_t vb = (int64_t)b * a.vstride;
    const T vx0 = a.vx[vb + (int64_t)i * g.ny + j], vxm = a.vx[vb + (int64_t)im1 * g.ny + j];
    const T vy0 = a.vy[vb + (int64_t)i * g.ny + j], vym = a.vy[vb + (int64_t)i * g.ny + jm1];
    const T fx0 = vx0 * (T(0.5) * (u00 + uxp)), fxm = vxm * (T(0.5) * (uxm + u00));
    const T fy0 = vy0 * (T(0.5) * (u00 + uyp)), fym = vym * (T(0.5) * (uym + u00));
    return -((fx0 - fxm) * a.rhx + (fy0 - fym) * a.rhy) +
           p.kappa * lap_at<T>(u00, uxp, uxm, uyp, uym, a.rhx2, a.rhy2);
  } else {
    const T ux2 = U(i2, j), uxm2 = U(im2, j), uy2 = U(i, j2), uym2 = U(i, jm2);
    const T upp = U(i1, j1), upm = U(i1, jm1), ump = U(im1, j1), umm = U(im1, jm1);
    const T kap = p.kappa;
    auto MU = [&](T c, T xp, T xm, T yp, T ym) -> T {
      return closure_generic<T>(a.mu, p.mu, c) - kap * lap_at<T>(c, xp, xm, yp, ym, a.rhx2, a.rhy2);
    };
    const T m00 = MU(u00, uxp, uxm, uyp, uym);
    const T mxp = MU(uxp, ux2, u00, upp, upm);
    const T mxm = MU(uxm, u00, uxm2, ump, umm);
    const T myp = MU(uyp, upp, ump, uy2, u00);
    const T mym = MU(uym, upm, umm, u00, uym2);
    const T d00 = closure_generic<T>(a.mob, p.mob, u00);
    const T dxp = closure_generic<T>(a.mob, p.mob, uxp);
    const T dxm = closure_generic<T>(a.mob, p.mob, uxm);
    const T dyp = closure_generic<T>(a.mob, p.mob, uyp);
    const T dym = closure_generic<T>(a.mob, p.mob, uym);
    const T fx0 = (T(0.5) * (d00 + dxp)) * ((mxp - m00) * a.rhx);
    const T fxm = (T(0.5) * (dxm + d00)) * ((m00 - mxm) * a.rhx);
    const T fy0 = (T(0.5) * (d00 + dyp)) * ((myp - m00) * a.rhy);
    const T fym = (T(0.5) * (dym + d00)) * ((m00 - mym) * a.rhy);
    return (fx0 - fxm) * a.rhx + (fy0 - fym) * a.rhy;
  }
}

template <typename T>
__device__ __forceinline__ void stage_update(const StageArgs<T>& a, int64_t idx, T k) {
  if (a.acc_mode == ACC_INIT) a.acc[idx] = a.y[idx] + a.b * k;
  if (a.out_mode == OUT_K) {
    a.out[idx] = k;
  } else if (a.out_mode == OUT_Y_PLUS_AK) {
    a.out[idx] = a.y[idx] + a.a * k;
  } else if (a.out_mode == OUT_ACC_PLUS_BK) {
    a.out[idx] = a.acc[idx] + a.b * k;
  } else if (a.out_mode == OUT_K_LC) {
    a.out[idx] = k;
    T r = a.y[idx];
    for (int j = 0; j < a.lc.n; ++j) r += a.lc.c[j] * a.lc.k[j][idx];
    a.lc.next[idx] = r + a.lc.c[a.lc.n] * k;
  }
  if (a.acc_mode == ACC_ADD) a.acc[idx] += a.b * k;
}

template <typename T, int EQ>
__global__ __launch_bounds__(256) void stage_generic_kernel(const StageArgs<T> a) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  const int b = blockIdx.z;
  if (i >= a.g.nx || j >= a.g.ny) return;
  const int64_t base = (int64_t)b * a.g.bstride + a.g.off;
  const EnvParams<T>& p = a.ep[b];
  T k = rhs_generic_point<T, EQ>(a, a.in + base, p, i, j, b);
  if (a.scaled) k *= p.kscale;
  stage_update<T>(a, base + (int64_t)i * a.g.ld + j, k);
}

// ---------------------------------------------------------------------------------------------------
// CahnHilliard2DSmoothedBoundary in TWO passes (cahn_hilliard.py:257-289): `inner` once per cell into a work
// field, then the psi-weighted flux divergence from its 5-point neighbourhood.  The one-pass generic kernel
// above re-evaluates `inner` -- a logit, two logs (mixing entropy), a square root and a psi-weighted Laplacian --
// at 5 points per cell; the same expressions evaluated once per cell give the same bits 3x faster
// (1024^2 fp32: 30.6 -> ~10 us per right-hand side).  Periodic layout (the SBM equations need it).
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sbm_inner_kernel(const StageArgs<T> a, T* __restrict__ inner_out) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  const int b = blockIdx.z;
  const int nx = a.g.nx, ny = a.g.ny;
  if (i >= nx || j >= ny) return;
  const EnvParams<T>& p = a.ep[b];
  const T* __restrict__ u = a.in + (int64_t)b * a.g.bstride;
  const int i1 = wrap_idx(i + 1, nx), im1 = wrap_idx(i - 1, nx), j1 = wrap_idx(j + 1, ny), jm1 = wrap_idx(j - 1, ny);
  auto U = [&](int ii, int jj) -> T { return u[(int64_t)ii * ny + jj]; };
  auto PS = [&](int ii, int jj) -> T { return a.psi[(int64_t)ii * ny + jj]; };
  const T c = U(i, j), xp = U(i1, j), xm = U(im1, j), yp = U(i, j1), ym = U(i, jm1);
  const T pc = PS(i, j), pxp = PS(i1, j), pxm = PS(im1, j), pyp = PS(i, j1), pym = PS(i, jm1);
  const T m = a.mask[(int64_t)i * ny + j];
  const T w = sqrt(p.kappa) * a.ngp[(int64_t)i * ny + j] * (a.tw_a * m + a.tw_b * (T(1) - m));
  // the expression of INNER in rhs_generic_point, term for term
  const T dx_hi = (T(0.5) * (pc + pxp)) * ((xp - c) * a.rhx), dx_lo = (T(0.5) * (pxm + pc)) * ((c - xm) * a.rhx);
  const T dy_hi = (T(0.5) * (pc + pyp)) * ((yp - c) * a.rhy), dy_lo = (T(0.5) * (pym + pc)) * ((c - ym) * a.rhy);
  const T lap = (dx_hi - dx_lo) * a.rhx + (dy_hi - dy_lo) * a.rhy;
  T r = closure_generic<T>(a.mu, p.mu, c) - (p.kappa / pc) * lap;
  r -= w * sqrt(T(2) * closure_generic<T>(a.fe, p.fe, c));
  inner_out[(int64_t)b * a.g.bstride + (int64_t)i * ny + j] = r;
}

template <typename T>
__global__ __launch_bounds__(256) void sbm_ch_stage_kernel(const StageArgs<T> a) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  const int b = blockIdx.z;
  const int nx = a.g.nx, ny = a.g.ny;
  if (i >= nx || j >= ny) return;
  const EnvParams<T>& p = a.ep[b];
  const int64_t base = (int64_t)b * a.g.bstride;
  const T* __restrict__ u = a.in + base;
  const T* __restrict__ in = a.mu3 + base;
  const int i1 = wrap_idx(i + 1, nx), im1 = wrap_idx(i - 1, nx), j1 = wrap_idx(j + 1, ny), jm1 = wrap_idx(j - 1, ny);
  auto at = [&](int ii, int jj) -> int64_t { return (int64_t)ii * ny + jj; };
  const T p00 = a.psi[at(i, j)], pxp = a.psi[at(i1, j)], pxm = a.psi[at(im1, j)], pyp = a.psi[at(i, j1)], pym = a.psi[at(i, jm1)];
  const T in00 = in[at(i, j)], inxp = in[at(i1, j)], inxm = in[at(im1, j)], inyp = in[at(i, j1)], inym = in[at(i, jm1)];
  const T d00 = closure_generic<T>(a.mob, p.mob, u[at(i, j)]);
  const T dxp = closure_generic<T>(a.mob, p.mob, u[at(i1, j)]), dxm = closure_generic<T>(a.mob, p.mob, u[at(im1, j)]);
  const T dyp = closure_generic<T>(a.mob, p.mob, u[at(i, j1)]), dym = closure_generic<T>(a.mob, p.mob, u[at(i, jm1)]);
  const T fx0 = (T(0.5) * (p00 + pxp)) * (T(0.5) * (d00 + dxp)) * ((inxp - in00) * a.rhx);
  const T fxm = (T(0.5) * (pxm + p00)) * (T(0.5) * (dxm + d00)) * ((in00 - inxm) * a.rhx);
  const T fy0 = (T(0.5) * (p00 + pyp)) * (T(0.5) * (d00 + dyp)) * ((inyp - in00) * a.rhy);
  const T fym = (T(0.5) * (pym + p00)) * (T(0.5) * (dym + d00)) * ((in00 - inym) * a.rhy);
  T k = ((fx0 - fxm) * a.rhx + (fy0 - fym) * a.rhy) / p00 + a.ngp[at(i, j)] * a.tsrc;
  if (a.scaled) k *= p.kscale;
  stage_update<T>(a, base + at(i, j), k);
}

// ---------------------------------------------------------------------------------------------------
// CahnHilliard3DPeriodic.rhs_fd (cahn_hilliard.py:180-200), fields [b][nx][ny][nz] with z contiguous.
// Two passes: mu = mu_h(u) - kappa lap7(u) into a work field, then k = div(D grad mu) from the 7-point
// neighbourhoods of mu and u with the stage update fused in.  (One thread per cell; the 32^3 - 64^3
// problems this class is used for upstream are launch-bound, not bandwidth-bound.)  CL: the closure class of the 2-D
// kernels (closures.hpp: fixed polynomial / logit forms unrolled; CL_GENERIC walks the family at run time -- with it the
// stage kernel spent 340 VALU instructions per cell, most of them in the seven mobility evaluations).
// ---------------------------------------------------------------------------------------------------
template <typename T, int CL>
__global__ __launch_bounds__(256) void ch3d_mu_kernel(const StageArgs<T> a, T* __restrict__ mu_out) {
  const int nx = a.g.nx, ny = a.g.ny, nz = a.g.nz;
  const int k = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int i = blockIdx.z % nx, b = blockIdx.z / nx;
  if (k >= nz || j >= ny) return;
  const T* __restrict__ u = a.in + (int64_t)b * a.g.bstride;
  const EnvParams<T>& p = a.ep[b];
  auto U = [&](int ii, int jj, int kk) -> T { return u[((int64_t)ii * ny + jj) * nz + kk]; };
  const int ip = (i + 1 == nx) ? 0 : i + 1, im = (i == 0) ? nx - 1 : i - 1;
  const int jp = (j + 1 == ny) ? 0 : j + 1, jm = (j == 0) ? ny - 1 : j - 1;
  const int kp = (k + 1 == nz) ? 0 : k + 1, km = (k == 0) ? nz - 1 : k - 1;
  const T c = U(i, j, k);
  const T lap = (U(ip, j, k) - T(2) * c + U(im, j, k)) * a.rhx2 + (U(i, jp, k) - T(2) * c + U(i, jm, k)) * a.rhy2 +
                (U(i, j, kp) - T(2) * c + U(i, j, km)) * a.rhz2;
  mu_out[(int64_t)b * a.g.bstride + ((int64_t)i * ny + j) * nz + k] = eval_mu<T, CL>(a.mu, p.mu, c) - p.kappa * lap;
}

template <typename T, int CL>
__global__ __launch_bounds__(256) void ch3d_stage_kernel(const StageArgs<T> a) {
  const int nx = a.g.nx, ny = a.g.ny, nz = a.g.nz;
  const int k = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int i = blockIdx.z % nx, b = blockIdx.z / nx;
  if (k >= nz || j >= ny) return;
  const int64_t base = (int64_t)b * a.g.bstride;
  const T* __restrict__ u = a.in + base;
  const T* __restrict__ m = a.mu3 + base;
  const EnvParams<T>& p = a.ep[b];
  auto at = [&](int ii, int jj, int kk) -> int64_t { return ((int64_t)ii * ny + jj) * nz + kk; };
  const int ip = (i + 1 == nx) ? 0 : i + 1, im = (i == 0) ? nx - 1 : i - 1;
  const int jp = (j + 1 == ny) ? 0 : j + 1, jm = (j == 0) ? ny - 1 : j - 1;
  const int kp = (k + 1 == nz) ? 0 : k + 1, km = (k == 0) ? nz - 1 : k - 1;
  const int64_t c0 = at(i, j, k);
  const T m0 = m[c0], d0 = eval_mob<T, CL>(a.mob, p.mob, u[c0]);
  // flux through the face between this cell and a neighbour: avg_face(D) * grad_face(mu)
  auto flux = [&](int64_t n, T rh, bool plus) -> T {
    const T dn = eval_mob<T, CL>(a.mob, p.mob, u[n]);
    const T g = plus ? (m[n] - m0) * rh : (m0 - m[n]) * rh;
    return (T(0.5) * (plus ? (d0 + dn) : (dn + d0))) * g;
  };
  const T kx = (flux(at(ip, j, k), a.rhx, true) - flux(at(im, j, k), a.rhx, false)) * a.rhx;
  const T ky = (flux(at(i, jp, k), a.rhy, true) - flux(at(i, jm, k), a.rhy, false)) * a.rhy;
  const T kz = (flux(at(i, j, kp), a.rhz, true) - flux(at(i, j, km), a.rhz, false)) * a.rhz;
  T kk = kx + ky + kz;
  if (a.scaled) kk *= p.kscale;
  stage_update<T>(a, base + c0, kk);
}

// (Also measured without effect: an XCD-aware block map -- the 1-D grid re-dealt so that an XCD owns a slab of consecutive
// x planes instead of every 8th block: 672 against 678-710 env-steps/s on 8 x 128^3, although the mu pass fetches 2.4 x
// the field in launch order.)
// (A single-pass LDS-brick form -- u on an 8 x 8 x 64 brick + 2 and mu on the brick + 1 in LDS, one launch per stage --
// was built and measured in round 4: bitwise equal, 249 us per stage of 8 x 128^3 against 146 us for the two passes above.
// Its 65 KB of LDS leave 8 waves per CU; the two simple kernels run 32 and find their neighbours in L1 / L2.  Removed.)

// jit.hip: closures compiled at run time (PDEOPT_CL_JIT; pdeopt_set_jit_closures)
bool jit_closures_active(const pdeopt_ctx* ctx);
template <typename T>
int launch_jit_stage(pdeopt_ctx* ctx, const StageArgs<T>& s);

}  // namespace pdeopt
