// N = 1024 as 32 x 32 (gfx950): every thread owns 32 points of a sequence, 32 threads (half a wave) serve it, a
// transform is TWO radix-32 butterflies in registers with ONE exchange through LDS between them -- against two
// exchanges for the 16 x 8 x 8 plan of fft_reg.hpp (each removed exchange was worth ~2.6 us per transform and
// 16-environment group on the IMEX passes).
//
//   n = j + 32 m (natural, thread j, register m)      k = q + 32 r (frequency, thread q, register r)
//   X[q + 32 r] = sum_j W32^(j r) [ W1024^(j q) sum_m W32^(m q) x[j + 32 m] ]
// i.e. radix-32 over the register index, twiddle W1024^(thread x register), transpose (thread <-> register)
// through LDS, radix-32 over the register index again.  The inverse has the same shape (the natural and the
// frequency layout are both "thread + 32 x register"), so dif and dit are one routine with the sign flipped,
// both global sides of a pass are coalesced (consecutive lanes hold consecutive elements), and
// FFT -> multiply -> IFFT chains through registers.
//
// LDS image of a sequence: 32 rows of 33 elements (row = register index of the writer, column = its thread):
// the writes of one register by 32 lanes are consecutive, the transposed reads walk the rows with stride 33
// elements -- conflict-free both ways.  32 threads are lanes of one wave: no s_barrier (the LDS unit executes a
// wave's instructions in order).
#pragma once

#include "fft_reg.hpp"

namespace pdeopt {

// y_q = sum_m v_m exp(SIGN 2 pi i q m / 32): two radix-16 butterflies on the even / odd inputs + the combine
template <typename T, int SIGN>
__device__ __forceinline__ void dft32(Cx<T>* v) {
  Cx<T> e[16], o[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    e[q] = v[2 * q];
    o[q] = v[2 * q + 1];
  }
  dft16<T, SIGN>(e);
  dft16<T, SIGN>(o);
  // w32^q = (cos(q pi / 16), SIGN sin(q pi / 16)), q = 0 .. 15
  constexpr double kc[16] = {1.0,
                             0.98078528040323044913,
                             0.92387953251128675613,
                             0.83146961230254523708,
                             0.70710678118654752440,
                             0.55557023301960222474,
                             0.38268343236508977173,
                             0.19509032201612826785,
                             0.0,
                             -0.19509032201612826785,
                             -0.38268343236508977173,
                             -0.55557023301960222474,
                             -0.70710678118654752440,
                             -0.83146961230254523708,
                             -0.92387953251128675613,
                             -0.98078528040323044913};
  constexpr double ks[16] = {0.0,
                             0.19509032201612826785,
                             0.38268343236508977173,
                             0.55557023301960222474,
                             0.70710678118654752440,
                             0.83146961230254523708,
                             0.92387953251128675613,
                             0.98078528040323044913,
                             1.0,
                             0.98078528040323044913,
                             0.92387953251128675613,
                             0.83146961230254523708,
                             0.70710678118654752440,
                             0.55557023301960222474,
                             0.38268343236508977173,
                             0.19509032201612826785};
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    Cx<T> t;
    if (q == 0)
      t = o[0];
    else if (q == 8)
      t = mul_si<T, SIGN>(o[8]);
    else
      t = cmul(o[q], Cx<T>{T(kc[q]), T(SIGN) * T(ks[q])});
    v[q] = cadd(e[q], t);
    v[q + 16] = csub(e[q], t);
  }
}

template <typename T, int PITCH = 33>
struct RegFft32x32 {
  using C = Cx<T>;
  static constexpr int N = 1024, PTS = 32, TT = 32;
  static constexpr int kPts = 32;
  static constexpr bool kWaveLocal = true;
  static constexpr int kPitch = PITCH;
  // the exchange moves the real and the imaginary parts in two rounds through the same scalar image: half the LDS
  // of a complex image (4.2 KB per fp32 sequence -> 4 workgroups of 8 sequences per CU instead of 2)
  using LdsT = T;
  static constexpr int NP = 32 * kPitch;  // LDS elements of one sequence's image
  static __device__ __forceinline__ int natural(int j, int m) { return j + 32 * m; }
  static __device__ __forceinline__ int freq(int j, int sl) { return j + 32 * sl; }

  // v[q] *= w^q, w = exp(SIGN 2 pi i j / 1024).  The powers w^1, w^2, w^4, w^8, w^16 come from the table (j 2^b <
  // 1024), every other one is its parent (q with the lowest set bit cleared) times one of them: at most 4 products
  // deep, and in numeric order of q only the parent chain (<= 5 values) is live.
  template <int SIGN>
  static __device__ __forceinline__ void twiddle_by_thread(C (&v)[32], const C* __restrict__ tw, int j) {
    C base[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) base[b] = twiddle<T, SIGN>(tw, j << b);
    C w[32];
#pragma unroll
    for (int q = 1; q < 32; ++q) {
      const int par = q & (q - 1), b = __builtin_ctz(q);
      w[q] = par == 0 ? base[b] : cmul(w[par], base[b]);
      v[q] = cmul(v[q], w[q]);
    }
  }

  // thread j's register q  ->  thread q's register j  (through the sequence's LDS image).  WL: the 32 threads of
  // the sequence are lanes of one wave (compiler barrier); otherwise workgroup barriers.
  template <bool WL = true>
  static __device__ __forceinline__ void transpose(C (&v)[32], T* __restrict__ seq, int j) {
#pragma unroll
    for (int q = 0; q < 32; ++q) seq[q * kPitch + j] = v[q].re;
    reg_fft_sync<WL>();
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q].re = seq[j * kPitch + q];
    reg_fft_sync<WL>();
#pragma unroll
    for (int q = 0; q < 32; ++q) seq[q * kPitch + j] = v[q].im;
    reg_fft_sync<WL>();
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q].im = seq[j * kPitch + q];
    reg_fft_sync<WL>();
  }

  // natural layout (thread j: x[j + 32 m]) -> frequency layout (thread j: X[j + 32 r]); SIGN -1 forward, +1
  // unnormalised inverse.  dit() is the same routine: frequency layout in, natural layout out.
  template <int SIGN, bool WL = true>
  static __device__ __forceinline__ void transform(C (&v)[32], T* __restrict__ seq, const C* __restrict__ tw, int j) {
    dft32<T, SIGN>(v);
    twiddle_by_thread<SIGN>(v, tw, j);
    transpose<WL>(v, seq, j);
    dft32<T, SIGN>(v);
  }
  template <int SIGN>
  static __device__ __forceinline__ void dif(C (&v)[32], T* __restrict__ seq, const C* __restrict__ tw, int j) {
    transform<SIGN>(v, seq, tw, j);
  }
  template <int SIGN>
  static __device__ __forceinline__ void dit(C (&v)[32], T* __restrict__ seq, const C* __restrict__ tw, int j) {
    transform<SIGN>(v, seq, tw, j);
  }
};

}  // namespace pdeopt
