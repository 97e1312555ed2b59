// Building blocks of the power-of-two FFTs (gfx950), radix 8 with a radix-4/2 tail: complex helpers,
// small DFTs in registers, the radix plan, the digit-reversal index maps and the padded LDS addressing.
// The transforms themselves are in fft_reg.hpp (butterflies in registers, exchanges through LDS).
//
// Why not rocFFT for the split-step: a Strang step is   ifft2(fft2(psi) E) -> pointwise -> ifft2(fft2(.) E)
// (pde_opt/numerics/solvers.py:99-122).  With a library FFT every arrow is a pass over HBM (and the
// strided column pass of rocFFT's 2-D plan runs at ~2 TB/s).  With the transform in LDS the
// pointwise operators fuse INTO the passes:  a column pass does FFT_x -> *E -> IFFT_x without leaving
// the CU, a row pass does IFFT_y -> *exp(b tau) (+ norm partial sums) -> FFT_y, so a whole step is
// 4 passes of 16 B/cell instead of 8 library passes + 4 pointwise kernels.
//
// Geometry shared by the transforms (in-place decimation-in-frequency positions):
//   dif: natural order in  -> position p holds X[rev(p)]          dit: the reverse
// rev() is the mixed-radix digit reversal of the radix list (8, 8, ..., tail); pos_of() its inverse.
// A pointwise spectral multiply between dif and dit therefore needs no reordering, and the global
// side of every pass stays in natural order (coalesced) -- only LDS addresses are permuted.
#pragma once

#include <hip/hip_runtime.h>

namespace pdeopt {

template <typename T>
struct Cx {
  T re, im;
};

template <typename T>
__device__ __forceinline__ Cx<T> cadd(Cx<T> a, Cx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T>
__device__ __forceinline__ Cx<T> csub(Cx<T> a, Cx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <typename T>
__device__ __forceinline__ Cx<T> cmul(Cx<T> a, Cx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
// multiply by  SIGN * i
template <typename T, int SIGN>
__device__ __forceinline__ Cx<T> mul_si(Cx<T> a) {
  return SIGN > 0 ? Cx<T>{-a.im, a.re} : Cx<T>{a.im, -a.re};
}

// ---- small DFTs in registers:  y_q = sum_m v_m exp(SIGN 2 pi i q m / R)
template <typename T, int SIGN>
__device__ __forceinline__ void dft2(Cx<T>* v) {
  const Cx<T> a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}
template <typename T, int SIGN>
__device__ __forceinline__ void dft4(Cx<T>& a0, Cx<T>& a1, Cx<T>& a2, Cx<T>& a3) {
  const Cx<T> t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_si<T, SIGN>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = cadd(t1, t3);
  a3 = csub(t1, t3);
}
template <typename T, int SIGN>
__device__ __forceinline__ void dft8(Cx<T>* v) {
  // even / odd DFT4, then the radix-2 combine with w8^q
  Cx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  Cx<T> o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4<T, SIGN>(e0, e1, e2, e3);
  dft4<T, SIGN>(o0, o1, o2, o3);
  const T h = T(0.70710678118654752440);
  // o1 *= (1 + s i)/sqrt2 ; o2 *= s i ; o3 *= (-1 + s i)/sqrt2
  const Cx<T> si1 = mul_si<T, SIGN>(o1), si3 = mul_si<T, SIGN>(o3);
  o1 = {(o1.re + si1.re) * h, (o1.im + si1.im) * h};
  o2 = mul_si<T, SIGN>(o2);
  o3 = {(si3.re - o3.re) * h, (si3.im - o3.im) * h};
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
template <typename T, int SIGN>
__device__ __forceinline__ void dft16(Cx<T>* v) {
  // even / odd DFT8, then the radix-2 combine with w16^q = (cos(q pi/8), SIGN sin(q pi/8))
  Cx<T> e[8], o[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    e[q] = v[2 * q];
    o[q] = v[2 * q + 1];
  }
  dft8<T, SIGN>(e);
  dft8<T, SIGN>(o);
  const T h = T(0.70710678118654752440), c1 = T(0.92387953251128673848), s1 = T(SIGN) * T(0.38268343236508978178);
  const T sc1 = T(SIGN) * c1, ss1 = T(SIGN) * s1;  // ss1 = sin(pi/8): s1 already carries SIGN
  o[1] = cmul(o[1], Cx<T>{c1, s1});
  o[3] = cmul(o[3], Cx<T>{ss1, sc1});
  o[5] = cmul(o[5], Cx<T>{-ss1, sc1});
  o[7] = cmul(o[7], Cx<T>{-c1, s1});
  const Cx<T> si2 = mul_si<T, SIGN>(o[2]), si6 = mul_si<T, SIGN>(o[6]);
  o[2] = {(o[2].re + si2.re) * h, (o[2].im + si2.im) * h};
  o[4] = mul_si<T, SIGN>(o[4]);
  o[6] = {(si6.re - o[6].re) * h, (si6.im - o[6].im) * h};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    v[q] = cadd(e[q], o[q]);
    v[q + 8] = csub(e[q], o[q]);
  }
}
template <typename T, int R, int SIGN>
__device__ __forceinline__ void dft_small(Cx<T>* v) {
  if constexpr (R == 16) dft16<T, SIGN>(v);
  if constexpr (R == 8) dft8<T, SIGN>(v);
  if constexpr (R == 4) dft4<T, SIGN>(v[0], v[1], v[2], v[3]);
  if constexpr (R == 2) dft2<T, SIGN>(v);
}

// ---- radix plan of a power of two: as many 8s as fit, then a tail of 4 or 2.  N = 1024 (threads own
// 16 points there, fft_reg.hpp) runs 16 x 8 x 8 instead of 8 x 8 x 8 x 2: one exchange through LDS
// less per transform.
#ifndef PDEOPT_FFT_RADIX16
#define PDEOPT_FFT_RADIX16 1
#endif
template <int N>
struct FftPlan {
  static constexpr int log2n() { int l = 0, n = N; while (n > 1) { n >>= 1; ++l; } return l; }
  static constexpr bool lead16 = PDEOPT_FFT_RADIX16 && N == 1024;
  static constexpr int n8 = lead16 ? (log2n() - 4) / 3 : log2n() / 3;
  static constexpr int tail = lead16 ? 1 : 1 << (log2n() % 3);  // 1, 2 or 4
  static constexpr int stages = lead16 ? n8 + 1 : n8 + (tail > 1 ? 1 : 0);
  static constexpr int radix(int i) { return lead16 ? (i == 0 ? 16 : 8) : (i < n8 ? 8 : tail); }
  // sub-length entering stage i (stage 0 sees N)
  static constexpr int sublen(int i) { int ns = N; for (int k = 0; k < i; ++k) ns /= radix(k); return ns; }
};

// position p (after dif)  ->  frequency index k
template <int N>
__device__ __forceinline__ int fft_rev(int p) {
  using P = FftPlan<N>;
  int k = 0, w = 1;
#pragma unroll
  for (int i = 0; i < P::stages; ++i) {
    const int R = P::radix(i), S = P::sublen(i) / R;
    k += ((p / S) % R) * w;
    w *= R;
  }
  return k;
}
// frequency / natural index k  ->  position p whose content dit() expects / dif() produces
template <int N>
__device__ __forceinline__ int fft_pos_of(int k) {
  using P = FftPlan<N>;
  int p = 0, w = 1;
#pragma unroll
  for (int i = 0; i < P::stages; ++i) {
    const int R = P::radix(i), S = P::sublen(i) / R;
    p += ((k / w) % R) * S;
    w *= R;
  }
  return p;
}

// LDS address of sequence position `pos`: one pad element per 8.  The LDS serves 16 lanes of a ds_read_b64 per
// clock from 32 four-byte banks, i.e. 16 eight-byte slots: a layout whose lanes stride 8 positions would put 16
// lanes on 2 slots; with the pad the stride is 9 slots, which walks all 16.  (Lane stride 1 pays for it -- 16
// consecutive positions span 17 slots, a 2-way conflict -- and lane stride 64 is not helped at all: RegFft keeps
// its last stage off that stride, see PDEOPT_FFT_LAST_IDENTITY; tools/lds_bank_model.py evaluates a layout.)
__device__ __forceinline__ constexpr int fft_lds_addr(int pos) { return pos + (pos >> 3); }
// padded length of one sequence (+1 so that consecutive sequences start one slot apart: the column
// pass loads/stores with the sequence index fastest across lanes)
template <int N>
constexpr int fft_lds_pitch() { return N + N / 8 + 1; }

// twiddle exp(SIGN 2 pi i n / N) from the table tw[n] = exp(-2 pi i n / N)
template <typename T, int SIGN>
__device__ __forceinline__ Cx<T> twiddle(const Cx<T>* __restrict__ tw, int n) {
  const Cx<T> w = tw[n];
  return SIGN < 0 ? w : Cx<T>{w.re, -w.im};
}

}  // namespace pdeopt
