// Register-resident power-of-two FFTs (gfx950): every thread owns PTS (8 or 16) points of one sequence,
// does the radix-8 (tail: radix-4 / radix-2) butterflies of a stage in registers, and only the
// re-grouping between two stages goes through LDS.
//
// Against fft_lds.hpp (data lives in LDS, every stage reads and writes it, a barrier after each):
//   * LDS traffic per transform drops from (stages + 1) read+write sweeps to (stages - 1);
//   * N / PTS threads serve one sequence, so with PTS = 8 (N <= 512) or 16 (N = 1024) a sequence lives
//     inside ONE wave and the exchanges need no s_barrier at all (LDS instructions of a wave execute
//     in order) -- the row passes of the split step become barrier-free;
//   * an inverse followed by a forward transform (the row pass of the Strang step, fft_lds.hpp header)
//     or a forward followed by an inverse (column pass) chain through registers: the stage where one
//     ends is the stage where the other starts, so a pass has 2 (stages - 1) exchanges in total.
//
// Same radix plan and index algebra as fft_lds.hpp (FftPlan, fft_rev, fft_pos_of, fft_lds_addr):
//   DIF stage i, butterfly bf in [0, N/R): blk = bf / S, jj = bf % S, S = sublen(i) / R,
//   positions blk * sublen(i) + jj + m S (m < R), outputs q > 0 times w^(jj q N / sublen(i)).
// Thread j of the TT = N/PTS threads of a sequence takes butterflies j + u TT (u < PTS/R).  In the LAST
// stage (S = 1) the assignment is by frequency instead: thread j holds the frequencies k = j + m TT,
// which are exactly PTS/R whole butterflies (a last-stage butterfly is {k_low + c N/R}), so the global
// side of a spectrum load / store is coalesced (consecutive k across lanes) with no reordering.
// In stage 0 thread j holds natural positions n = natural(j, m) = j + (multiple of TT): coalesced too.
#pragma once

#include "fft_lds.hpp"

namespace pdeopt {

// ordering point of an exchange: compiler barrier for a wave-local sequence (the LDS unit serves one
// wave's instructions in order), workgroup barrier when a sequence spans waves
template <bool WAVE_LOCAL>
__device__ __forceinline__ void reg_fft_sync() {
  if constexpr (WAVE_LOCAL)
    asm volatile("" ::: "memory");
  else
    __syncthreads();
}

template <int N>
constexpr int reg_default_pts() { return N > 512 ? 16 : 8; }

// Which positions a thread owns in the LAST stage.  1 (default): thread j owns positions (j + u TT) R .. + R - 1,
// so the lanes of a wave walk the image with stride R and the one-pad-per-8 addressing puts every 16 lanes of a
// ds_read_b64 / ds_write_b64 (the LDS serves 16 lanes x 8 bytes per clock from 32 banks) on 16 different bank
// pairs.  0: the positions whose frequencies are j + slot TT (consecutive across lanes) -- lane stride 64 (512:
// 8 8 8) or 128 positions, 4- to 8-way conflicts: SQ_LDS_BANK_CONFLICT was 50 % of the LDS cycles of the Strang
// passes (profiles/pmc_r02.json), which tools/lds_bank_model.py reproduces to the digit.  The frequencies of a
// wave stay one contiguous block either way (digit-reversed WITHIN it), so global accesses coalesce as before.
#ifndef PDEOPT_FFT_LAST_IDENTITY
#define PDEOPT_FFT_LAST_IDENTITY 1
#endif
template <typename T, int N, int PTS = reg_default_pts<N>()>
struct RegFft {
  using P = FftPlan<N>;
  using C = Cx<T>;
  static constexpr int TT = N / PTS;  // threads per sequence
  static constexpr int L = P::stages;
  static constexpr bool kWaveLocal = TT <= 64;
  static constexpr int kPts = PTS;
  using LdsT = C;                                   // element type of a sequence's LDS image
  static constexpr int NP = fft_lds_pitch<N>();     // and its length
  static_assert(PTS == 8 || PTS == 16, "8 or 16 points per thread");
  static_assert(PTS % P::radix(0) == 0, "a thread owns whole butterflies");

  // position (in the in-place DIF geometry) of register slot `slot` of thread j at stage STAGE
  template <int STAGE>
  static __device__ __forceinline__ int pos(int j, int slot) {
    constexpr int R = P::radix(STAGE), Ns = P::sublen(STAGE), S = Ns / R;
    const int u = slot / R, m = slot % R;
    if constexpr (S == 1) {
#if PDEOPT_FFT_LAST_IDENTITY
      return (j + u * TT) * R + m;  // thread-major: the frequencies land digit-reversed across the lanes (freq())
#else
      return fft_pos_of<N>(j + u * TT) + m;  // k = j + (u + m PER) TT: m is the top digit of k, the bottom digit of pos
#endif
    } else {
      const int bf = j + u * TT;
      const int blk = bf / S, jj = bf % S;
      return blk * Ns + jj + m * S;
    }
  }
  // natural index held in `slot` at stage 0 (real-space side of a pass)
  static __device__ __forceinline__ int natural(int j, int slot) { return pos<0>(j, slot); }
  // frequency held in `slot` after the last DIF stage / expected there before the first DIT stage
  static __device__ __forceinline__ int freq(int j, int slot) {
#if PDEOPT_FFT_LAST_IDENTITY
    return fft_rev<N>(pos<L - 1>(j, slot));
#else
    constexpr int R = P::radix(L - 1), PER = PTS / R;
    return j + (slot / R + (slot % R) * PER) * TT;
#endif
  }

  // LDS address of slot (u, m) = base(u) + a compile-time constant: within one butterfly the positions
  // are p0 + m S and  fft_lds_addr(p0 + m S) = fft_lds_addr(p0) + fft_lds_addr(m S)  for every stage of
  // the plans (S a multiple of 8, or p0 mod 8 < S with S | 8, or -- last stage -- p0 a multiple of R):
  // a thread keeps PTS/R base addresses per layout instead of PTS full ones and the offsets fold into
  // the ds_read / ds_write immediates.
  template <int STAGE>
  static __device__ __forceinline__ int base(int j, int u) {
    return fft_lds_addr(pos<STAGE>(j, u * P::radix(STAGE)));
  }
  template <int STAGE>
  static constexpr int off(int m) {
    return fft_lds_addr(m * (P::sublen(STAGE) / P::radix(STAGE)));
  }

  // butterflies (+ twiddles) of one stage on the registers of thread j
  template <int STAGE, int SIGN, bool DIT>
  static __device__ __forceinline__ void stage(C (&v)[PTS], const C* __restrict__ tw, int j) {
    constexpr int R = P::radix(STAGE), Ns = P::sublen(STAGE), S = Ns / R, PER = PTS / R;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      C* b = v + u * R;
      C w[R > 8 ? 9 : R];
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
        if constexpr (R > 8) w[8] = cmul(w[4], w[4]);
      }
      // radix 16: w^9 .. w^15 = w^8 w^(q-8) are formed where they are used, never held together
      auto wq = [&](const int q) -> C {
        if constexpr (R > 8)
          return q <= 8 ? w[q] : cmul(w[8], w[q > 8 ? q - 8 : 0]);
        else
          return w[q];
      };
      if constexpr (S > 1 && DIT) {
#pragma unroll
        for (int q = 1; q < R; ++q) b[q] = cmul(b[q], wq(q));
      }
      dft_small<T, R, SIGN>(b);
      if constexpr (S > 1 && !DIT) {
#pragma unroll
        for (int q = 1; q < R; ++q) b[q] = cmul(b[q], wq(q));
      }
      // one butterfly at a time: interleaving the independent butterflies of a 16-point thread only
      // raises the register pressure (spills at 1024-thread workgroups otherwise)
      if constexpr (PER > 1 && PTS > 8) __builtin_amdgcn_sched_barrier(0);
    }
  }

  // the two halves of a re-grouping through the sequence's LDS image: put() stores the registers in
  // the slot layout of stage ST, get() loads them in the slot layout of stage ST
  template <int ST>
  static __device__ __forceinline__ void put(const C (&v)[PTS], C* __restrict__ seq, int j) {
    constexpr int R = P::radix(ST);
#pragma unroll
    for (int u = 0; u < PTS / R; ++u) {
      C* const b = seq + base<ST>(j, u);
#pragma unroll
      for (int m = 0; m < R; ++m) b[off<ST>(m)] = v[u * R + m];
    }
  }
  template <int ST>
  static __device__ __forceinline__ void get(C (&v)[PTS], const C* __restrict__ seq, int j) {
    constexpr int R = P::radix(ST);
#pragma unroll
    for (int u = 0; u < PTS / R; ++u) {
      const C* const b = seq + base<ST>(j, u);
#pragma unroll
      for (int m = 0; m < R; ++m) v[u * R + m] = b[off<ST>(m)];
    }
  }
  // re-group from the slot layout of stage FROM to that of stage TO
  // (WL: the threads of the sequence are lanes of one wave)
  template <int FROM, int TO, bool WL>
  static __device__ __forceinline__ void exchange(C (&v)[PTS], C* __restrict__ seq, int j) {
    put<FROM>(v, seq, j);
    reg_fft_sync<WL>();
    get<TO>(v, seq, j);
    reg_fft_sync<WL>();  // the next exchange overwrites the image
  }

  // DIF geometry from the stage-0 layout (natural positions) to the last-stage layout (frequencies
  // freq()); SIGN = -1 forward, +1 unnormalised inverse
  template <int SIGN, bool WL = kWaveLocal, int ST = 0>
  static __device__ __forceinline__ void dif(C (&v)[PTS], C* __restrict__ seq, const C* __restrict__ tw, int j) {
    stage<ST, SIGN, false>(v, tw, j);
    if constexpr (ST + 1 < L) {
      exchange<ST, ST + 1, WL>(v, seq, j);
      dif<SIGN, WL, ST + 1>(v, seq, tw, j);
    }
  }
  // DIT from the last-stage layout (frequencies freq()) to the stage-0 layout (natural positions)
  template <int SIGN, bool WL = kWaveLocal, int ST = L - 1>
  static __device__ __forceinline__ void dit(C (&v)[PTS], C* __restrict__ seq, const C* __restrict__ tw, int j) {
    stage<ST, SIGN, true>(v, tw, j);
    if constexpr (ST > 0) {
      exchange<ST, ST - 1, WL>(v, seq, j);
      dit<SIGN, WL, ST - 1>(v, seq, tw, j);
    }
  }

  // The same transforms for a workgroup whose GLOBAL side wants one thread layout (thread jo of a
  // sequence anywhere in the workgroup, e.g. the sequence index fastest across lanes for a strided
  // pass) while stages 1 .. L-1 run with the TT threads of a sequence inside one wave (thread ji):
  // stage 0 only needs the registers, so the exchange between stages 0 and 1 is where the layout
  // changes and the only place that needs a workgroup barrier -- one per transform, because after it
  // an image is touched by nobody but the wave that owns the sequence.
  // (seqo / seqi: the image of the sequence the thread belongs to in the outer / inner layout.)
  template <int SIGN>
  static __device__ __forceinline__ void dif_split(C (&v)[PTS], C* __restrict__ seqo, int jo, C* __restrict__ seqi,
                                                   int ji, const C* __restrict__ tw) {
    static_assert(L >= 2 && kWaveLocal, "needs an exchange and wave-sized sequences");
    stage<0, SIGN, false>(v, tw, jo);
    put<0>(v, seqo, jo);
    __syncthreads();
    get<1>(v, seqi, ji);
    reg_fft_sync<true>();
    dif<SIGN, true, 1>(v, seqi, tw, ji);
  }
  template <int SIGN, int ST = L - 1>
  static __device__ __forceinline__ void dit_inner(C (&v)[PTS], C* __restrict__ seq, const C* __restrict__ tw, int j) {
    stage<ST, SIGN, true>(v, tw, j);
    if constexpr (ST > 1) {
      exchange<ST, ST - 1, true>(v, seq, j);
      dit_inner<SIGN, ST - 1>(v, seq, tw, j);
    }
  }
  template <int SIGN>
  static __device__ __forceinline__ void dit_split(C (&v)[PTS], C* __restrict__ seqi, int ji, C* __restrict__ seqo,
                                                   int jo, const C* __restrict__ tw) {
    static_assert(L >= 2 && kWaveLocal, "needs an exchange and wave-sized sequences");
    dit_inner<SIGN>(v, seqi, tw, ji);
    put<1>(v, seqi, ji);
    __syncthreads();
    get<0>(v, seqo, jo);
    stage<0, SIGN, true>(v, tw, jo);
  }
};

}  // namespace pdeopt
