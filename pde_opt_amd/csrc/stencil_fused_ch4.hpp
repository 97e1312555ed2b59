// A whole classical RK4 substep of Cahn-Hilliard in ONE pass over HBM (fp32, periodic, divisible grids): the four
// radius-2 stages are chained inside a workgroup, so a substep reads y once (tile + 8 halo) and writes y' once --
// 2 words per cell (+ the halo re-reads, L2 hits) instead of the 7 of the stage-pair kernels (stencil_fused.hpp),
// which with two groups side by side run at 0.66 of the HBM peak / ~0.87 of what streaming kernels reach.
//
//   mu1 = mu(y)   on tile+7    k1 = div(D grad mu1) on tile+6    w2 = y + dt/2 k1   (LDS B, tile+6)
//   mu2 = mu(w2)  on tile+5    k2                   on tile+4    w3 = y + dt/2 k2   (LDS A, in place over y, tile+4)
//   mu3 = mu(w3)  on tile+3    k3                   on tile+2    w4 = y + dt   k3   (LDS B, tile+2)
//   mu4 = mu(w4)  on tile+1    k4                   on tile      y' = y + dt/6 (k1 + 2 k2 + 2 k3 + k4)
//   cahn_hilliard.py:89-109, derivatives.py:8-61
//
// Three LDS arrays -- A (y, then w3), M (mu of the stage), B (w2, then w4) -- 48 + 46 + 44 rows of 36 vectors =
// 79.5 KB: two 512-thread workgroups per CU (4 waves per SIMD) against three for the pair kernels.  The stage
// inputs ALTERNATE between A and B, so the stage's writes need no barrier of their own (nobody reads the array
// being written: stencil_fused_ac4.hpp's scheme); 9 barriers per tile and substep against the pair kernels' 10.
// Every stage evaluates k on the thread's own micro-tile (2 rows x 1 vector, the marching form of the pair kernels)
// and on the ring of the region the later stages still need (560 / 336 / 200 vectors).  y, the base of every w, is
// in registers for the own cells; ring cells read it from A (stage 2 overwrites A with w3 cell by cell, each cell
// by the thread that read its y) except stage 3's, fetched up front.
// Redundant work against the cell count: mu x1.62 / 1.48 / 1.26 / 1.13, k x1.55 / 1.33 / 1.2 / 1 -- 15 % more than
// the two pair kernels do.  The arithmetic IS the pair kernels' (same mu form, face fluxes, divergence, update
// association): results are bitwise theirs.
//
// Measured on the headline (32 x 1024^2, two groups side by side; profiles/r03_ch4_experiments.txt):
//   512 threads (every thread owns cells, rings first)                   1880-1905 env-steps/s   (stage pairs: 1826-1880)
//   1024 threads = 512 owners + 512 helpers (mu passes, tile load)       1966
//   ... helpers take the rings WHILE the owners march (this kernel)      2034
//   ... as persistent workgroups, next tile + parameters by LDS-DMA under stage 4   1843  (tile load by DMA alone: 1946)
// Phase ablation of a 78.6 us launch (PDEOPT_CH4_ABLATE): no mu passes 50.8, no rings 68.6, no marches 62.4, neither 46.9,
// none of the three (load, 9 barriers, w writes, store) 23.3 -- close to a sum: two workgroups per CU overlap little, and
// a persistent workgroup hiding its own load does not change that (third measurement of that kind on this part: the
// stage-pair kernel in round 2, the Allen-Cahn kernel and this one in round 3).
#pragma once

#include <atomic>
#include <type_traits>

#include "stencil_fused.hpp"

namespace pdeopt {

#ifndef PDEOPT_CH_QUAD_FUSE
#define PDEOPT_CH_QUAD_FUSE 2  // the PDEOPT_OPT_FUSE_STAGES value that selects this kernel
#endif

template <typename T>
struct Quad4Args {
  const T* y;  // state (read: tile + 8)
  T* out;      // y' (a different buffer: neighbouring tiles still read y)
  T h2, h3, h6, dt;
  T rhx, rhy;    // 0.5 / hx^2, 0.5 / hy^2 (face_flux)
  T rhx2, rhy2;  // 1 / hx^2, 1 / hy^2
  Geo g;
  const EnvParams<T>* ep;
  ClosureSpec mu, mob;
  // A rank's tile of a decomposed field (halo-8 layout, HALO kernels; stencil_fused.hpp: PairArgs has the same
  // fields for the stage pairs).  The tile + 8 input of this kernel IS the layout's halo: one kernel and one exchange
  // per substep, nothing re-evaluated on a ring outside the tile.
  //   nbase[0] != nullptr: the edge tiles take the halo cells of y straight from the neighbour ranks' strips (nbase[q]:
  //   the strip of neighbour q in {up, down, left, right, UL, UR, DL, DR} -- a slice of the gathered buffer, or the
  //   neighbour's own buffer mapped into this process) -- no unpack launch; nullptr: the halo frame of the field holds
  //   them (pdeopt_halo_unpack ran).
  //   strip != nullptr: the edge tiles also write the cells of y' within 8 of the tile border into this rank's strip
  //   (halo.hip's layout) -- no pack launch.
  const T* nbase[8];
  T* strip;
  int64_t strip_env;
};

#ifndef PDEOPT_CH4_THREADS
#define PDEOPT_CH4_THREADS 1024
#endif
// LPR lanes (= 16-byte vectors) across a tile row.  32: the pair kernels' 32 x 128 tile.  16: a 64 x 64 tile -- the same 512
// owners, but a shorter perimeter in vectors: the rings are 496 / 272 / 200 vectors instead of 560 / 336 / 200 (stage 1's
// ring fits the 512 helpers in ONE trip), the tile + 8 halo is x1.56 instead of x1.69 of the tile, 74.9 KB of LDS -- and
// 5.5 % SLOWER on the headline (short rows); it serves the grids that 64 x 64 divides and 32 x 128 does not.
template <int LPR_, int NOWN_ = 512, int NT_ = PDEOPT_CH4_THREADS, int WGS_PER_CU_ = 2>
struct Ch4GeomT {
  // NOWN threads own the tile's cells (2 rows x 1 vector each); threads past them, if any, are helpers: they take their
  // share of the mu passes, the rings and the tile load and sit out the marches
  static constexpr int LPR = LPR_;
  static constexpr int NOWN = NOWN_, NT = NT_, V = 4, RPT = 2, HV = 2;
  static constexpr int kWavesPerSimd = (WGS_PER_CU_ * NT / 64 + 3) / 4;  // WGS_PER_CU workgroups per CU (LDS)
  static constexpr int TX = (NOWN / LPR) * RPT;         // 32 / 64 rows
  static constexpr int PV = LPR + 2 * HV;               // vectors per LDS row: the tile + 8 columns each side
  static constexpr int P = PV * V, TY = LPR * V;
  static constexpr int kRowsA = TX + 16, kRowsM = TX + 14, kRowsB = TX + 12;
  static constexpr size_t lds_bytes() { return (size_t)((kRowsA + kRowsM + kRowsB) * P + 4 * V) * sizeof(float); }
  // the region tile + H minus the tile, in vectors: 2 H full rows of NCV(H) vectors + 2 HVS(H) side vectors per tile row
  static constexpr int hvs(int H) { return (H + V - 1) / V; }
  static constexpr int ncv(int H) { return LPR + 2 * hvs(H); }
  static constexpr int ring(int H) { return 2 * H * ncv(H) + TX * 2 * hvs(H); }
};
using Ch4Geom = Ch4GeomT<32>;

template <int CL, typename G, bool HALO = false>
__global__ __launch_bounds__(G::NT, G::kWavesPerSimd) void ch_rk4_quad_kernel(const Quad4Args<float> a, const int tiles_i, const int tiles_j,
                                                                    const int nblk, const int xcd_remap) {
  using T = float;
  using Vec = typename VecOf<T>::type;
  constexpr int V = G::V, RPT = G::RPT, TX = G::TX, PV = G::PV, P = G::P, TY = G::TY, NT = G::NT, HV = G::HV, LPR = G::LPR;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // pointers to (tile row 0, LDS vector column 0) of each array: element (r, cv) = base + r P + cv V, r may be negative
  T* const A0 = reinterpret_cast<T*>(smem_raw) + V + 8 * P;
  T* const M0 = A0 + (G::kRowsA - 8) * P + V + 7 * P;
  T* const B0 = M0 + (G::kRowsM - 7) * P + V + 6 * P;

  int ti, tj, b;
  decode_tile(blockIdx.x, tiles_i, tiles_j, nblk, xcd_remap, &ti, &tj, &b);
  const int i0 = ti * TX;
  const int j0 = tj * TY;

  const Geo& g = a.g;
  const int64_t ld = g.ld;
  const int64_t base = (int64_t)b * g.bstride + g.off;
  const EnvParams<T>& p = a.ep[b];
  const T* __restrict__ in = a.y + base;
  const T kap = p.kappa;

  const int tid = threadIdx.x;
  const int lx = tid & (LPR - 1);
  const int ly = tid / LPR;
  const int r0 = ly * RPT;
  const int cvo = lx + HV;
  const bool owner = tid < G::NOWN;  // wave-uniform

  // ---- y on tile + 8 -> A, one tile row per wave and trip (stencil_generic.hpp: load_rows_per_wave)
  bool edge_tile = false;  // wave-uniform
  if constexpr (HALO) edge_tile = ti == 0 || ti == tiles_i - 1 || tj == 0 || tj == tiles_j - 1;
  if (HALO && edge_tile && a.nbase[0] != nullptr) {
    // Fused unpack (stage_pair_kernel does the same for its tile + 4 launch): interior cells from the field, the cells
    // of the 8-wide halo from the strip piece of the neighbour they belong to (halo.hip: my halo piece q <- piece
    // FROM[q] of neighbour q).  A vector never straddles two sources (every boundary is a multiple of V).  Per thread:
    // one source pointer + pitch for each row class, chosen by its column class; per trip the wave-uniform row picks.
    constexpr int NW = NT / 64, H = 8;
    static_assert(HV * V == H, "the tile + 8 input is the layout's halo");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if (lane < PV) {
      const int nx = g.nx, ny = g.ny;
      const StripOffsets<H> so(nx, ny);
      const int64_t eoff = (int64_t)b * a.strip_env;  // this environment inside a rank's strip
      const int gj = j0 - H + lane * V;
      const T *pT, *pM, *pB;
      int64_t qT, qM, qB;  // pitches
      if (gj >= 0 && gj < ny) {
        pT = a.nbase[0] + eoff + so.bottom + gj; qT = ny;  // up's bottom rows
        pM = in + gj; qM = ld;
        pB = a.nbase[1] + eoff + so.top + gj; qB = ny;     // down's top rows
      } else if (gj < 0) {
        const int c = gj + H;
        pT = a.nbase[4] + eoff + so.br + c; pM = a.nbase[2] + eoff + so.right + c; pB = a.nbase[6] + eoff + so.tr + c;
        qT = qM = qB = H;
      } else {
        const int c = gj - ny;
        pT = a.nbase[5] + eoff + so.bl + c; pM = a.nbase[3] + eoff + so.left + c; pB = a.nbase[7] + eoff + so.tl + c;
        qT = qM = qB = H;
      }
      T* const lds = A0 - 8 * P + lane * V;
      constexpr int kRows = G::kRowsA, kTrips = (kRows + NW - 1) / NW;
      Vec f[kTrips];
#pragma unroll
      for (int k = 0; k < kTrips; ++k) {
        int row = wave + k * NW;
        if constexpr (kRows % NW != 0) row = row < kRows ? row : kRows - 1;
        const int gi = i0 - H + row;  // wave-uniform
        const T* src;
        if (gi < 0) src = pT + (int64_t)(gi + H) * qT;
        else if (gi < nx) src = pM + (int64_t)gi * qM;
        else src = pB + (int64_t)(gi - nx) * qB;
        f[k] = *reinterpret_cast<const Vec*>(src);
      }
#pragma unroll
      for (int k = 0; k < kTrips; ++k) {
        const int row = wave + k * NW;
        if (kRows % NW == 0 || k + 1 < kTrips || row < kRows) *reinterpret_cast<Vec*>(lds + row * P) = f[k];
      }
    }
  } else {
    // periodic field: wrap by index; a rank's padded tile: the neighbours are in memory (halo frame)
    auto wrap_row = [&](int gi) { return HALO ? gi : tile_wrap(gi, g.nx, false); };
    auto wrap_col = [&](int gj) { return HALO ? gj : tile_wrap(gj, g.ny, false); };
    load_rows_per_wave<T, V, PV, NT, G::kRowsA, Vec>(A0 - 8 * P, P, in, ld, i0 - 8, j0 - HV * V, wrap_row, wrap_col, tid);
  }
  __syncthreads();

  constexpr bool FOLD_MU = PDEOPT_PAIR_FOLD_MU && CL == CL_LOGIT1;
  T fA = T(0), fB = T(0), q1 = T(0);
  if constexpr (FOLD_MU) {
    fA = -kap * a.rhx2;
    fB = -kap * a.rhy2;
    q1 = p.mu[1] - T(2) * (fA + fB);
  }

  // mu(src) -> M on the tile + H region: rows [-H, TX + H), the NCV vectors of ncv(H) (stage_pair_kernel: mu_pass)
  auto mu_pass = [&](auto h_c, const T* const src0) {
    constexpr int H = decltype(h_c)::value;
#if defined(PDEOPT_CH4_ABLATE) && (PDEOPT_CH4_ABLATE & 1)  // TIMING ONLY (tools/mkvariant.sh): no mu passes
    return;
#endif
    constexpr int NCV = G::ncv(H), CV0 = HV - G::hvs(H), NVEC = (TX + 2 * H) * NCV;
#pragma unroll 3
    for (int base0 = 0; base0 < NVEC; base0 += NT) {
      const int idx = base0 + tid;
      if (idx >= NVEC) break;
      const int rr = idx / NCV;
      const int cv = CV0 + (idx - rr * NCV);
      const int r = rr - H;
      const T* c_ = src0 + r * P + cv * V;
      const Vec c = *reinterpret_cast<const Vec*>(c_);
      const Vec xp = *reinterpret_cast<const Vec*>(c_ + P);
      const Vec xm = *reinterpret_cast<const Vec*>(c_ - P);
      const T left = nb_left<T, Vec, V>(c_), right = nb_right<T, Vec, V>(c_);
      Vec m;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T ym = (e == 0) ? left : c[e - 1];
        const T yp = (e == V - 1) ? right : c[e + 1];
        if constexpr (FOLD_MU)
          m[e] = fA * (xp[e] + xm[e]) + (fB * (yp + ym) + (q1 * c[e] + p.mu[0] + t_logit<T>(c[e])));
        else
          m[e] = eval_mu<T, CL>(a.mu, p.mu, c[e]) - kap * lap_at<T>(c[e], xp[e], xm[e], yp, ym, a.rhx2, a.rhy2);
      }
      *reinterpret_cast<Vec*>(M0 + r * P + cv * V) = m;
    }
  };

  // k at one vector (tile row r, LDS vector column cv) of the stage whose input is in src0 (stage_pair_kernel: k_at)
  auto k_at = [&](const T* const src0, const int r, const int cv) -> Vec {
    const T* mp = M0 + r * P + cv * V;
    const T* up = src0 + r * P + cv * V;
    return flux_divergence<T, CL, Vec, V>(a.mob, p.mob, *reinterpret_cast<const Vec*>(mp - P), *reinterpret_cast<const Vec*>(mp),
                                          *reinterpret_cast<const Vec*>(mp + P), *reinterpret_cast<const Vec*>(up - P),
                                          *reinterpret_cast<const Vec*>(up), *reinterpret_cast<const Vec*>(up + P),
                                          nb_left<T, Vec, V>(mp), nb_right<T, Vec, V>(mp), nb_left<T, Vec, V>(up),
                                          nb_right<T, Vec, V>(up), a.rhx, a.rhy);
  };
  // k on the own micro-tile, marching down the rows (stage_pair_kernel: march)
  auto march = [&](const T* const src0, Vec* kout) {
    const T* mp = M0 + (r0 - 1) * P + cvo * V;
    const T* up = src0 + (r0 - 1) * P + cvo * V;
    Vec m_lo = *reinterpret_cast<const Vec*>(mp);
    Vec d_lo = mob_vec<T, CL, Vec, V>(a.mob, p.mob, *reinterpret_cast<const Vec*>(up));
    mp += P;
    up += P;
    Vec m_c = *reinterpret_cast<const Vec*>(mp);
    Vec u_c = *reinterpret_cast<const Vec*>(up);
    Vec d_c = mob_vec<T, CL, Vec, V>(a.mob, p.mob, u_c);
    Vec fx_lo;
#pragma unroll
    for (int e = 0; e < V; ++e) fx_lo[e] = face_flux<T>(d_lo[e], d_c[e], m_lo[e], m_c[e], a.rhx);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const Vec m_hi = *reinterpret_cast<const Vec*>(mp + P);
      const Vec u_hi = *reinterpret_cast<const Vec*>(up + P);
      const Vec d_hi = mob_vec<T, CL, Vec, V>(a.mob, p.mob, u_hi);
      const T ml = nb_left<T, Vec, V>(mp), mr = nb_right<T, Vec, V>(mp);
      const T dl = eval_mob<T, CL>(a.mob, p.mob, nb_left<T, Vec, V>(up)), dr = eval_mob<T, CL>(a.mob, p.mob, nb_right<T, Vec, V>(up));
      const Vec dy = div_y<T, Vec, V>(m_c, d_c, ml, mr, dl, dr, a.rhy);
      Vec fx_hi, k;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        fx_hi[e] = face_flux<T>(d_c[e], d_hi[e], m_c[e], m_hi[e], a.rhx);
        k[e] = div_sum<T>(fx_hi[e] - fx_lo[e], a.rhx, dy[e]);
      }
      kout[r] = k;
      m_c = m_hi;
      d_c = d_hi;
      fx_lo = fx_hi;
      mp += P;
      up += P;
    }
  };
  // ring vector idx of the region tile + H: tile row / LDS vector column
  auto ring_coord = [&](auto h_c, const int idx, int* r, int* cv) {
    constexpr int H = decltype(h_c)::value;
    constexpr int NCV = G::ncv(H), HVS = G::hvs(H), CV0 = HV - HVS, TOP = 2 * H * NCV;
    if (idx < TOP) {
      const int q = idx / NCV;
      *r = (q < H) ? (q - H) : (TX + q - H);
      *cv = CV0 + (idx - q * NCV);
    } else {
      const int t2 = idx - TOP;
      const int rr = t2 / (2 * HVS), s = t2 - rr * (2 * HVS);
      *r = rr;
      *cv = s < HVS ? (CV0 + s) : (HV + LPR + (s - HVS));
    }
  };

  Vec yown[RPT], acc[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    acc[r] = Vec{};
    yown[r] = owner ? *reinterpret_cast<const Vec*>(A0 + (r0 + r) * P + cvo * V) : Vec{};
  }
  // Ring work: with helper threads (NT > NOWN) the helpers take the rings while the owners march -- different waves,
  // the same phase, so a stage's k pass costs max(march, ring) instead of their sum; without helpers every thread
  // takes ring vectors first (the non-marching flux_divergence is the register peak) and then marches.
  constexpr bool HELPERS = NT > G::NOWN;
  constexpr int kRingWorkers = HELPERS ? NT - G::NOWN : NT;
  const int rw = HELPERS ? tid - G::NOWN : tid;  // ring worker index (negative: not a ring worker)
  // y at this thread's ring vector of stage 3 (tile + 2 ring): stage 2 runs w3 over it in A
  static_assert(G::ring(2) <= kRingWorkers, "stage 3's ring is one trip");
  Vec yring3 = Vec{};
  if (rw >= 0 && rw < G::ring(2)) {
    int r3, c3;
    ring_coord(std::integral_constant<int, 2>{}, rw, &r3, &c3);
    yring3 = *reinterpret_cast<const Vec*>(A0 + r3 * P + c3 * V);
  }

  // One of the stages 1..3 after its mu pass: k on the ring of tile + H and on the own cells; w = y + cw k into dst0.
  auto stage = [&](auto h_c, const T* const src0, T* const dst0, const T cw, const T bw) {
    constexpr int H = decltype(h_c)::value;
#if defined(PDEOPT_CH4_ABLATE) && (PDEOPT_CH4_ABLATE & 2)  // TIMING ONLY: no ring work
    if (false) {
#else
    if (rw >= 0) {
#endif
      for (int idx = rw; idx < G::ring(H); idx += kRingWorkers) {
        int rr, rc;
        ring_coord(h_c, idx, &rr, &rc);
        const Vec k = k_at(src0, rr, rc);
        // y of the ring cell: still in A for stages 1 and 2 (stage 2's dst IS A: read, then overwritten by this thread)
        const Vec yr = H == 2 ? yring3 : *reinterpret_cast<const Vec*>(A0 + rr * P + rc * V);
        *reinterpret_cast<Vec*>(dst0 + rr * P + rc * V) = yr + cw * k;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (owner) {
      Vec k[RPT];
#if defined(PDEOPT_CH4_ABLATE) && (PDEOPT_CH4_ABLATE & 4)  // TIMING ONLY: no march
      for (int r = 0; r < RPT; ++r) k[r] = *reinterpret_cast<const Vec*>(M0 + (r0 + r) * P + cvo * V);
#else
      march(src0, k);
#endif
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        *reinterpret_cast<Vec*>(dst0 + (r0 + r) * P + cvo * V) = yown[r] + cw * k[r];
        if constexpr (H == 6) acc[r] = yown[r] + bw * k[r];
        else acc[r] = acc[r] + bw * k[r];
      }
    }
  };

  mu_pass(std::integral_constant<int, 7>{}, A0);
  __syncthreads();
  stage(std::integral_constant<int, 6>{}, A0, B0, a.h2, a.h6);  // k1 on tile+6, w2 -> B
  __syncthreads();
  mu_pass(std::integral_constant<int, 5>{}, B0);
  __syncthreads();
  stage(std::integral_constant<int, 4>{}, B0, A0, a.h2, a.h3);  // k2 on tile+4, w3 -> A (over y)
  __syncthreads();
  mu_pass(std::integral_constant<int, 3>{}, A0);
  __syncthreads();
  stage(std::integral_constant<int, 2>{}, A0, B0, a.dt, a.h3);  // k3 on tile+2, w4 -> B
  __syncthreads();
  mu_pass(std::integral_constant<int, 1>{}, B0);
  __syncthreads();

  // ---- stage 4 on the tile, combine, store
  if (owner) {
    Vec k4[RPT];
#if defined(PDEOPT_CH4_ABLATE) && (PDEOPT_CH4_ABLATE & 4)
    for (int r = 0; r < RPT; ++r) k4[r] = *reinterpret_cast<const Vec*>(M0 + (r0 + r) * P + cvo * V);
#else
    march(B0, k4);
#endif
    const int64_t pidx0 = base + (int64_t)(i0 + r0) * ld + (j0 + lx * V);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const Vec ynew = acc[r] + a.h6 * k4[r];
      *reinterpret_cast<Vec*>(a.out + pidx0 + r * ld) = ynew;
      if constexpr (HALO) {
        // fused pack (stage_pair_kernel's PAIR_34 epilogue): wave-uniform test first, interior tiles skip everything
        if (a.strip != nullptr && edge_tile) {
          constexpr int H = 8;
          const StripOffsets<H> so(g.nx, g.ny);
          T* const st = a.strip + (int64_t)b * a.strip_env;
          const int gi = i0 + r0 + r, gj = j0 + lx * V;
          const bool top = gi < H, bot = gi >= g.nx - H, lef = gj < H, rig = gj >= g.ny - H;
          const int bi = gi - (g.nx - H), rj = gj - (g.ny - H);
          if (top) *reinterpret_cast<Vec*>(st + so.top + (int64_t)gi * g.ny + gj) = ynew;
          if (bot) *reinterpret_cast<Vec*>(st + so.bottom + (int64_t)bi * g.ny + gj) = ynew;
          if (lef) *reinterpret_cast<Vec*>(st + so.left + (int64_t)gi * H + gj) = ynew;
          if (rig) *reinterpret_cast<Vec*>(st + so.right + (int64_t)gi * H + rj) = ynew;
          if (top && lef) *reinterpret_cast<Vec*>(st + so.tl + gi * H + gj) = ynew;
          if (top && rig) *reinterpret_cast<Vec*>(st + so.tr + gi * H + rj) = ynew;
          if (bot && lef) *reinterpret_cast<Vec*>(st + so.bl + bi * H + gj) = ynew;
          if (bot && rig) *reinterpret_cast<Vec*>(st + so.br + bi * H + rj) = ynew;
        }
      }
    }
  }
}

// whether the single-pass Cahn-Hilliard RK4 kernel covers the configured problem (the default where it applies; PDEOPT_OPT_FUSE_STAGES = 1 keeps the stage pairs)
#ifndef PDEOPT_CH4_TILE
#define PDEOPT_CH4_TILE 32  // preferred tile: 32 (x 128); 64 (x 64) where only that divides the grid.  Measured on the headline: 32 x 128 2040, 64 x 64 1927 env-steps/s
                            // (less redundant work, but 320-byte rows: 20 of 64 lanes per load, shorter bursts)
#endif
// which tile shape runs the configured grid: 0 = none (the kernel does not apply), else the tile's rows (32 or 64)
inline int ch_quad_tile(const pdeopt_ctx* ctx) {
  const pdeopt_problem& p = ctx->prob;
  if (p.equation != PDEOPT_EQ_CAHN_HILLIARD || p.dtype != PDEOPT_F32 || p.derivs != PDEOPT_DERIVS_FD) return 0;
  // periodic fields, and a rank's tile of a decomposed field in the halo-8 layout (its 8-cell halo is this kernel's tile + 8 input)
  if ((ctx->halo != 0 && ctx->halo != 8) || ctx->opt_kernel_path == 1 || ctx->opt_debug_ablate) return 0;
  if (ctx->opt_tile_rows == 16) return 0;  // a caller asking for 16-row tiles gets the pair kernels
  // (tiled_supported() also asks a padded layout for the pair kernels' 32-vector rows; this kernel's 64 x 64 tile has 16)
  if (!ctx->halo && !tiled_supported<float>(ctx)) return 0;
  if (classify_closures(p.mu, p.mob) == CL_GENERIC) return 0;
  // divisible grids only; the tile + 8 halo wraps at most once
  const bool ok64 = p.nx % 64 == 0 && p.ny % 64 == 0, ok32 = p.nx % 32 == 0 && p.ny % 128 == 0;
  if (ctx->opt_tile_rows == 64) return ok64 ? 64 : 0;
  if (ctx->opt_tile_rows == 32) return ok32 ? 32 : 0;
  if (PDEOPT_CH4_TILE == 64 && ok64) return 64;
  return ok32 ? 32 : (ok64 ? 64 : 0);
}
// whether the single-pass Cahn-Hilliard RK4 kernel covers the configured problem (the default where it applies;
// PDEOPT_OPT_FUSE_STAGES = 1 keeps the stage pairs)
inline bool ch_quad_supported(const pdeopt_ctx* ctx) { return ch_quad_tile(ctx) != 0; }
// ... and does an RK4 substep of the configured problem run on it?
inline bool ch_quad_chosen(const pdeopt_ctx* ctx) {
  return (ctx->opt_fuse_stages == 0 || ctx->opt_fuse_stages == PDEOPT_CH_QUAD_FUSE) && ch_quad_supported(ctx);
}

template <typename G>
int launch_ch_quad_g(pdeopt_ctx* ctx, const void* y, void* out, double dt) {
  const pdeopt_problem& p = ctx->prob;
  Quad4Args<float> s{};
  s.g = make_geo(ctx);
  const int64_t woff = (int64_t)ctx->win_lo * s.g.bstride;
  s.y = static_cast<const float*>(y) + woff;
  s.out = static_cast<float*>(out) + woff;
  s.dt = (float)dt; s.h2 = (float)(dt / 2); s.h3 = (float)(dt / 3); s.h6 = (float)(dt / 6);
  s.rhx = (float)(0.5 / (p.hx * p.hx)); s.rhy = (float)(0.5 / (p.hy * p.hy));
  s.rhx2 = (float)(1.0 / (p.hx * p.hx)); s.rhy2 = (float)(1.0 / (p.hy * p.hy));
  s.ep = static_cast<const EnvParams<float>*>(ctx->env_params_dev) + ctx->win_lo;
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  const int tiles_i = p.nx / G::TX, tiles_j = p.ny / G::TY;
  const int64_t nblk64 = (int64_t)tiles_i * tiles_j * ctx->win_n;
  if (nblk64 > 0x7fffffffLL) return fail(ctx, PDEOPT_EINVAL, "too many tiles");
  const int nblk = (int)nblk64;
  ctx->n_stage_launches++;
  const int cl = classify_closures(p.mu, p.mob);
  const int remap = tile_flags(nblk, tiles_i, tiles_j);
  const size_t lds = G::lds_bytes();
  auto go = [&](auto kern, const char* name) -> int {
    // 75-80 KB of dynamic LDS need the opt-in once per device and kernel (a static per instantiation of this lambda)
    static std::atomic<uint64_t> allowed{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (!(allowed.load(std::memory_order_relaxed) & bit)) {
      PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      allowed.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(G::NT), lds, ctx->stream, s, tiles_i, tiles_j, nblk, remap);
    PDEOPT_HIP_CHECK(ctx, hipGetLastError());
    ctx->last_kernel = std::string(name) + (G::TX == 64 ? ",rows64>" : ",rows32>");
    return PDEOPT_OK;
  };
  if (ctx->halo == 8) {
    // decomposed field: halo cells from the gathered strips / the new strip out of the store epilogue (rk4_substep_h8)
    s.strip = static_cast<float*>(ctx->pair_strip);
    s.strip_env = 2LL * 8 * p.ny + 2LL * p.nx * 8 + 4LL * 64;
    fill_neighbour_strips<float>(ctx, s.strip_env * p.batch, s.nbase);
    if (cl == CL_LOGIT && p.mu.n <= 2) return go(ch_rk4_quad_kernel<CL_LOGIT1, G, true>, "rk4_quad<f32,CH,halo8,logit");
    if (cl == CL_LOGIT) return go(ch_rk4_quad_kernel<CL_LOGIT, G, true>, "rk4_quad<f32,CH,halo8,logit");
    return go(ch_rk4_quad_kernel<CL_POLY, G, true>, "rk4_quad<f32,CH,halo8,poly");
  }
  if (cl == CL_LOGIT && p.mu.n <= 2) return go(ch_rk4_quad_kernel<CL_LOGIT1, G>, "rk4_quad<f32,CH,logit");
  if (cl == CL_LOGIT) return go(ch_rk4_quad_kernel<CL_LOGIT, G>, "rk4_quad<f32,CH,logit");
  return go(ch_rk4_quad_kernel<CL_POLY, G>, "rk4_quad<f32,CH,poly");
}

inline int launch_ch_quad(pdeopt_ctx* ctx, const void* y, void* out, double dt) {
#ifdef PDEOPT_CH4_ROWS16  // A/B build: 16 x 128 tiles, 256 owners + helpers, three workgroups per CU (51.8 KB each)
  if (ch_quad_tile(ctx) == 32) return launch_ch_quad_g<Ch4GeomT<32, 256, PDEOPT_CH4_ROWS16, 3>>(ctx, y, out, dt);
#endif
  return ch_quad_tile(ctx) == 64 ? launch_ch_quad_g<Ch4GeomT<16>>(ctx, y, out, dt) : launch_ch_quad_g<Ch4GeomT<32>>(ctx, y, out, dt);
}

}  // namespace pdeopt
