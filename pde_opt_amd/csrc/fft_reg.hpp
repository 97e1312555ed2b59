// Register-resident power-of-two FFTs (gfx950): every thread owns 8 points of one sequence, does the
// radix-8 (tail: radix-4 / radix-2) butterflies of a stage in registers, and only the re-grouping
// between two stages goes through LDS.
//
// Against fft_lds.hpp (data lives in LDS, every stage reads and writes it, a barrier after each):
//   * LDS traffic per transform drops from (stages + 1) read+write sweeps to (stages - 1);
//   * N / 8 threads serve one sequence, so for N <= 512 a sequence lives inside ONE wave and the
//     exchanges need no s_barrier at all (LDS instructions of a wave execute in order) -- the row
//     passes of the split step become barrier-free;
//   * an inverse followed by a forward transform (the row pass of the Strang step, fft_lds.hpp header)
//     or a forward followed by an inverse (column pass) chain through registers: the stage where one
//     ends is the stage where the other starts, so a pass has 2 (stages - 1) exchanges in total.
//
// Same radix plan and index algebra as fft_lds.hpp (FftPlan, fft_rev, fft_pos_of, fft_lds_addr):
//   DIF stage i, butterfly bf in [0, N/R): blk = bf / S, jj = bf % S, S = sublen(i) / R,
//   positions blk * sublen(i) + jj + m S (m < R), outputs q > 0 times w^(jj q N / sublen(i)).
// Thread j of the N/8 threads of a sequence takes butterflies j + u N/8 (u < 8/R).  In the LAST stage
// (S = 1) the assignment is by frequency instead: thread j holds the frequencies k = j + m N/8, which
// are exactly 8/R whole butterflies (a last-stage butterfly is {k_low + c N/R}), so the global side of
// a spectrum load / store is coalesced (64 consecutive k per wave instruction) with no reordering.
// In stage 0 thread j holds the natural positions n = j + m N/8: coalesced on the real-space side.
#pragma once

#include "fft_lds.hpp"

namespace pdeopt {

template <int N>
struct RegFft {
  using P = FftPlan<N>;
  static constexpr int TT = N / 8;  // threads per sequence
  static constexpr int L = P::stages;
  static constexpr bool kWaveLocal = TT <= 64;
};

// ordering point of an exchange: compiler barrier for a wave-local sequence (the LDS unit serves one
// wave's instructions in order), workgroup barrier when a sequence spans waves
template <bool WAVE_LOCAL>
__device__ __forceinline__ void reg_fft_sync() {
  if constexpr (WAVE_LOCAL)
    asm volatile("" ::: "memory");
  else
    __syncthreads();
}

// position (in the in-place DIF geometry) of register slot `slot` of thread j at stage STAGE
template <int N, int STAGE>
__device__ __forceinline__ int reg_pos(int j, int slot) {
  using P = FftPlan<N>;
  constexpr int R = P::radix(STAGE), Ns = P::sublen(STAGE), S = Ns / R, TT = N / 8;
  const int u = slot / R, m = slot % R;
  if constexpr (S == 1) {
    return fft_pos_of<N>(j + u * TT) + m;  // k = j + (u + m PER) TT: m is the top digit of k, the bottom digit of pos
  } else {
    const int bf = j + u * TT;
    const int blk = bf / S, jj = bf % S;
    return blk * Ns + jj + m * S;
  }
}
// frequency held in `slot` after the last DIF stage / expected there before the first DIT stage
template <int N>
__device__ __forceinline__ int reg_freq(int j, int slot) {
  using P = FftPlan<N>;
  constexpr int R = P::radix(P::stages - 1), PER = 8 / R, TT = N / 8;
  return j + (slot / R + (slot % R) * PER) * TT;
}

// butterflies (+ twiddles) of one stage on the 8 registers of thread j
template <typename T, int N, int STAGE, int SIGN, bool DIT>
__device__ __forceinline__ void reg_stage(Cx<T> (&v)[8], const Cx<T>* __restrict__ tw, int j) {
  using P = FftPlan<N>;
  constexpr int R = P::radix(STAGE), Ns = P::sublen(STAGE), S = Ns / R, PER = 8 / R, TT = N / 8;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    Cx<T>* b = v + u * R;
    Cx<T> w[R];
    if constexpr (S > 1) {
      const int jj = (j + u * TT) % S;
      // powers of the base twiddle by multiplication: one table read instead of R - 1
      w[1] = twiddle<T, SIGN>(tw, jj * (N / Ns));
      if constexpr (R > 2) {
        w[2] = cmul(w[1], w[1]);
        w[3] = cmul(w[2], w[1]);
      }
      if constexpr (R > 4) {
        w[4] = cmul(w[2], w[2]);
        w[5] = cmul(w[4], w[1]);
        w[6] = cmul(w[3], w[3]);
        w[7] = cmul(w[4], w[3]);
      }
      if constexpr (DIT) {
#pragma unroll
        for (int q = 1; q < R; ++q) b[q] = cmul(b[q], w[q]);
      }
    }
    dft_small<T, R, SIGN>(b);
    if constexpr (S > 1 && !DIT) {
#pragma unroll
      for (int q = 1; q < R; ++q) b[q] = cmul(b[q], w[q]);
    }
  }
}

// re-group from the slot layout of stage FROM to that of stage TO through the sequence's LDS image
// (WL: the N/8 threads of the sequence are lanes of one wave)
template <typename T, int N, int FROM, int TO, bool WL>
__device__ __forceinline__ void reg_exchange(Cx<T> (&v)[8], Cx<T>* __restrict__ seq, int j) {
#pragma unroll
  for (int s = 0; s < 8; ++s) seq[fft_lds_addr(reg_pos<N, FROM>(j, s))] = v[s];
  reg_fft_sync<WL>();
#pragma unroll
  for (int s = 0; s < 8; ++s) v[s] = seq[fft_lds_addr(reg_pos<N, TO>(j, s))];
  reg_fft_sync<WL>();  // the next exchange overwrites the image
}

// forward-geometry (DIF) transform from stage 0 layout (natural positions) to last-stage layout
// (frequencies reg_freq); SIGN = -1 forward, +1 unnormalised inverse
template <typename T, int N, int SIGN, bool WL = RegFft<N>::kWaveLocal, int ST = 0>
__device__ __forceinline__ void reg_fft_dif(Cx<T> (&v)[8], Cx<T>* __restrict__ seq, const Cx<T>* __restrict__ tw,
                                            int j) {
  constexpr int L = RegFft<N>::L;
  reg_stage<T, N, ST, SIGN, false>(v, tw, j);
  if constexpr (ST + 1 < L) {
    reg_exchange<T, N, ST, ST + 1, WL>(v, seq, j);
    reg_fft_dif<T, N, SIGN, WL, ST + 1>(v, seq, tw, j);
  }
}
// DIT transform from last-stage layout (frequencies reg_freq) to stage 0 layout (natural positions)
template <typename T, int N, int SIGN, bool WL = RegFft<N>::kWaveLocal, int ST = RegFft<N>::L - 1>
__device__ __forceinline__ void reg_fft_dit(Cx<T> (&v)[8], Cx<T>* __restrict__ seq, const Cx<T>* __restrict__ tw,
                                            int j) {
  reg_stage<T, N, ST, SIGN, true>(v, tw, j);
  if constexpr (ST > 0) {
    reg_exchange<T, N, ST, ST - 1, WL>(v, seq, j);
    reg_fft_dit<T, N, SIGN, WL, ST - 1>(v, seq, tw, j);
  }
}

}  // namespace pdeopt
