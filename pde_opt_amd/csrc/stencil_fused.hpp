// Temporally fused Runge-Kutta stage PAIRS for Cahn-Hilliard (gfx950).
//
// The per-stage kernel (stencil_tiled.hpp) moves the compulsory 16 words/cell of a classical RK4
// substep (3+5+5+3).  Two consecutive stages can share one pass over HBM if the
// intermediate stage input never leaves the chip: a workgroup loads its tile with a 4-cell halo,
// evaluates stage A on the tile + 2 ring, keeps  w = base + a_A k_A  in LDS only, and evaluates
// stage B on the tile from it.
//
//   PAIR_12:  k1 = f(y);  w = y + dt/2 k1;  k2 = f(w)
//             TB  = y + dt/2 k2                       reads y            (1 word)
//             ACC = y + dt/6 k1 + dt/3 k2             writes TB, ACC     (2 words)
//   PAIR_34:  k3 = f(TB); w = y + dt k3;    k4 = f(w)
//             Y'  = ACC + dt/3 k3 + dt/6 k4           reads TB, y, ACC   (3 words)
//             (Y' is a different buffer: neighbours still read y on their ring)  writes Y' (1 word)
//
// = 7 words/cell/substep instead of 16, at the price of re-evaluating stage A on the 2-cell ring
// (x1.33 at 16-row tiles, x1.2 at 32-row tiles) -- the halo re-reads are L2 hits (XCD-aware map).
// Same arithmetic as the per-stage kernels (cahn_hilliard.py:89-109); the ring
// and the interior go through one flux routine so a cell's value does not depend on which
// workgroup computed it.
//
// Phases per tile (barrier between each):
//   P1 stage-A input on tile+4            HBM -> LDS sU        (16-byte loads, wrap by index)
//   P2 mu_A on tile+3                     sU  -> LDS sMu       (one closure evaluation per point)
//   P3 k_A on own micro-tile + one ring vector per thread      -> registers
//   P4 w on tile+2                        registers -> sU (in place)
//   P5 mu_B on tile+1                     sU  -> sMu
//   P6 k_B on own micro-tile, stage updates, 16-byte stores    -> HBM
#pragma once

#include "stencil_tiled.hpp"

namespace pdeopt {

// wave-uniform value -> scalar register (the compiler cannot prove uniformity through the tile decode)
__device__ __forceinline__ int uniform_i(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float uniform_f(float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
__device__ __forceinline__ double uniform_f(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

enum { PAIR_12 = 0, PAIR_34 = 1, PAIR_K = 2 };  // PAIR_K: one stage, out = f(in) (the IMEX slope)

// The scalar neighbours left / right of a thread's 16-byte vector.  Read as ds_read_b32 their lane stride is 4 dwords:
// the 32 lanes of an access group fall on 8 banks, a 4-way conflict = 8 LDS cycles per wave-instruction, and they are
// a third of this kernel's LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.36).  PDEOPT_WIDE_NEIGHBOURS reads
// the whole neighbouring VECTOR instead (ds_read_b128: 4 cycles, conflict-free) and keeps one element; the empty asm
// stops the compiler from narrowing the load back to the one dword that is used.
#ifndef PDEOPT_WIDE_NEIGHBOURS
#define PDEOPT_WIDE_NEIGHBOURS 0
#endif
// PDEOPT_NB_ABLATE (TIMING ONLY, tools/mkvariant.sh; results are wrong): 1 = no neighbour read at all (an element of the
// centre vector stands in), 2 = the same number of ds_read_b32, but from addresses one dword apart across the lanes
// (conflict-free): what the 4-way bank conflict of the real reads costs, and what removing the reads would buy
#ifndef PDEOPT_NB_ABLATE
#define PDEOPT_NB_ABLATE 0
#endif
template <typename T, typename Vec, int V>
__device__ __forceinline__ T nb_left(const T* c_) {
#if PDEOPT_NB_ABLATE == 1
  return c_[0];
#elif PDEOPT_NB_ABLATE == 2
  return c_[-1 - (int)((3u * threadIdx.x) & 31u)];
#elif PDEOPT_WIDE_NEIGHBOURS
  Vec t = *reinterpret_cast<const Vec*>(c_ - V);
  asm volatile("" : "+v"(t));
  return t[V - 1];
#else
  return c_[-1];
#endif
}
template <typename T, typename Vec, int V>
__device__ __forceinline__ T nb_right(const T* c_) {
#if PDEOPT_NB_ABLATE == 1
  return c_[V - 1];
#elif PDEOPT_NB_ABLATE == 2
  return c_[V - (int)((3u * threadIdx.x) & 31u)];
#elif PDEOPT_WIDE_NEIGHBOURS
  Vec t = *reinterpret_cast<const Vec*>(c_ + V);
  asm volatile("" : "+v"(t));
  return t[0];
#else
  return c_[V];
#endif
}

// Phase-ablation hooks (TIMING ONLY, tools/ablate_pair.sh) are compiled in with -DPDEOPT_PAIR_ABLATE
// (tools/mkvariant.sh): even as never-taken uniform branches they change hipcc's schedule of the
// product kernel (96 -> 80 VGPRs, 3 % slower), so the shipped build does not carry them.
#ifdef PDEOPT_PAIR_ABLATE
#define PDEOPT_ABL(a, bit) ((a).dbg & (bit))
#else
#define PDEOPT_ABL(a, bit) false
#endif

template <typename T>
struct PairArgs {
  const T* in;    // stage-A input (PAIR_12: y;  PAIR_34: TB)
  const T* y;     // substep base state (PAIR_34 only; PAIR_12 takes it from the tile)
  const T* acc;   // PAIR_34: running accumulator (read)
  T* out;         // PAIR_12: TB;  PAIR_34: Y'
  T* acc_out;     // PAIR_12: ACC (write)
  T aA, bA;       // w = base + aA kA ;  accp = (y | acc) + bA kA
  T aB, bB;       // PAIR_12: out = y + aB kB, acc_out = accp + bB kB ;  PAIR_34: out = accp + bB kB
  T rhx, rhy;    // 0.5 / hx^2, 0.5 / hy^2 (folded flux constants, see face_flux)
  T rhx2, rhy2;  // 1 / hx^2, 1 / hy^2 (Laplacian)
  Geo g;
  const EnvParams<T>* ep;
  ClosureSpec mu, mob;
  int dbg;  // TIMING-ONLY ablation bits (PDEOPT_OPT_DEBUG_ABLATE): 1 skip mu passes, 2 skip marches, 4 skip ring,
            // 8 skip the tile loads, 16 skip the stores
  int part;  // which tiles this launch computes: 0 all, 1 interior tiles only, 2 edge tiles only (tiles in the
             // first / last tile row or column: the ones that read halo cells of a decomposed field, so the
             // interior can run while the halo exchange is in flight)
  // PAIR_34 of a decomposed field (halo-8 layout): the edge tiles also write the cells of the new state within 8
  // of the tile border into the rank's halo strip -- csrc/halo.hip's layout, [top 8 rows][bottom 8 rows]
  // [left 8 cols][right 8 cols][TL][TR][BL][BR] per environment -- so no pack launch is needed (nullptr: off)
  T* strip;
  int64_t strip_env;  // strip elements per environment
  // PAIR_12 of a decomposed field (halo-8 layout): the launch covers the tile + `ext` ring (pointers pre-shifted by
  // -ext rows / columns).  With nbase[0] != nullptr the edge tiles take the halo cells of their input straight from the
  // neighbour ranks' strips (nbase[q]: the strip of neighbour q in {up, down, left, right, UL, UR, DL, DR}) instead of
  // from the field's halo frame, and write the frame cells they own back into the field (the second stage pair reads
  // y there): no unpack launch.
  const T* nbase[8];  // strips of {up, down, left, right, UL, UR, DL, DR}: rank-major slices of the gathered buffer, or the
                      // neighbours' own strip buffers mapped into this process (peer-mapped exchange, comm.hip)
  int ext;
};

// offsets of the 8 pieces of a halo strip of width H for an nx x ny tile (halo.hip: decode())
template <int H>
struct StripOffsets {
  int64_t top, bottom, left, right, tl, tr, bl, br;
  __device__ __forceinline__ StripOffsets(int nx, int ny) {
    const int64_t rows = (int64_t)H * ny, cols = (int64_t)nx * H, cor = H * H;
    top = 0; bottom = rows; left = 2 * rows; right = 2 * rows + cols;
    tl = 2 * rows + 2 * cols; tr = tl + cor; bl = tr + cor; br = bl + cor;
  }
};

// does this launch skip tile (ti, tj)?  (uniform per workgroup; evaluated before any barrier)
__device__ __forceinline__ bool tile_skipped(int part, int ti, int tj, int tiles_i, int tiles_j) {
  if (part == 0) return false;
  const bool edge = ti == 0 || ti == tiles_i - 1 || tj == 0 || tj == tiles_j - 1;
  return (part == 1) == edge;
}

// flux through the face between cells a and b (b = a + 1 along the axis with spacing 1/rh):
// avg_face(D) * grad_face(mu)  (derivatives.py:24-31,39-46; cahn_hilliard.py:105-106).  Every face
// value in this file goes through this one expression so that a face shared by two rows, two
// threads or two workgroups evaluates to the same bits.
//
// The constants are folded: flux = (D_a + D_b)(mu_b - mu_a) here, and the 1/2 of the face average and
// the 1/h of the face gradient join the 1/h of the divergence in ONE factor 0.5/h^2 per axis
// (PairArgs::rhx / rhy), applied to the difference of two face values: 2 multiplies per face less
// than the literal form, ~5 % of this VALU-bound kernel.
//
// Rounding must not depend on WHERE in a workgroup tile a cell falls: a decomposed field tiles the plane differently
// from the monolithic run (and the halo-8 layout shifts the tiling by 4 cells), and both must give the same bits.
// Under -ffp-contract=fast-honor-pragmas (hipcc's default) the difference of two face fluxes, a*b - c*d, may be
// contracted either way round -- fma(a, b, -(c*d)) or fma(-c, d, a*b) -- and the backend picks by use counts, which
// differ between the marching form (a face flux is shared by two rows) and the ring form (it is not): seen as 1 ulp
// on 0.02-0.06 % of the cells.  PDEOPT_FLUX_PIN: 0 nothing pinned; 1 the FMA of k = dFx rhx + dFy rhy spelled out;
// 2 = 1 + every face-flux product rounded (contraction off inside face_flux), so the differences are plain
// subtractions; 3 (default) = the face-flux products AND the product dFy rhy rounded, k left to the compiler --
// its only contraction is then fma(dFx, rhx, .).  Same-box A/B on the headline (tools/ab_pin.sh, interleaved x 3) and
// the decomposed == monolithic tests of tests/test_gpu_decomp.py on 27 layouts:
//     0: 1763 env-steps/s, 8 layouts differ   1: 1684, 4 differ   2: 1660, bitwise   3: 1743 (-1.1 %), bitwise
#ifndef PDEOPT_FLUX_PIN
#define PDEOPT_FLUX_PIN 3
#endif
template <typename T>
__device__ __forceinline__ T face_flux(T d_a, T d_b, T m_a, T m_b, T /*unused*/) {
#if PDEOPT_FLUX_PIN >= 2
#pragma clang fp contract(off)
#endif
  return (d_a + d_b) * (m_b - m_a);
}
// k = dfx * rhx + dy with dy = dFy * rhy already formed: one FMA, the x term fused
template <typename T>
__device__ __forceinline__ T div_sum(T dfx, T rhx, T dy) {
#if PDEOPT_FLUX_PIN == 1 || PDEOPT_FLUX_PIN == 2
  return __builtin_fma(dfx, rhx, dy);
#else
  return dfx * rhx + dy;
#endif
}

// y-divergence of the face fluxes of one vector of cells (needs the scalar neighbours left / right)
template <typename T, typename Vec, int V>
__device__ __forceinline__ Vec div_y(Vec m_c, Vec d_c, T ml, T mr, T dl, T dr, T rhy) {
  T fy[V + 1];
  fy[0] = face_flux<T>(dl, d_c[0], ml, m_c[0], rhy);
#pragma unroll
  for (int e = 1; e < V; ++e) fy[e] = face_flux<T>(d_c[e - 1], d_c[e], m_c[e - 1], m_c[e], rhy);
  fy[V] = face_flux<T>(d_c[V - 1], dr, m_c[V - 1], mr, rhy);
  Vec r;
  {
#if PDEOPT_FLUX_PIN >= 3
#pragma clang fp contract(off)  // the product stays a product: k = dfx rhx + r then has ONE contraction (div_sum)
#endif
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = (fy[e + 1] - fy[e]) * rhy;
  }
  return r;
}

template <typename T, int CL, typename Vec, int V>
__device__ __forceinline__ Vec mob_vec(const ClosureSpec& ms, const T* __restrict__ mcoef, Vec u) {
  Vec d;
#pragma unroll
  for (int e = 0; e < V; ++e) d[e] = eval_mob<T, CL>(ms, mcoef, u[e]);
  return d;
}

// k = div(D grad mu) for one vector of cells, given the three mu / u rows around it and the scalar
// neighbours left and right of the centre row (derivatives.py:54-61, cahn_hilliard.py:109)
template <typename T, int CL, typename Vec, int V>
__device__ __forceinline__ Vec flux_divergence(const ClosureSpec& ms, const T* __restrict__ mcoef,
                                               Vec m_dn, Vec m_c, Vec m_up, Vec u_dn, Vec u_c, Vec u_up,
                                               T ml, T mr, T ul, T ur, T rhx, T rhy) {
  const Vec d_dn = mob_vec<T, CL, Vec, V>(ms, mcoef, u_dn);
  const Vec d_c = mob_vec<T, CL, Vec, V>(ms, mcoef, u_c);
  const Vec d_up = mob_vec<T, CL, Vec, V>(ms, mcoef, u_up);
  const T dl = eval_mob<T, CL>(ms, mcoef, ul), dr = eval_mob<T, CL>(ms, mcoef, ur);
  const Vec dy = div_y<T, Vec, V>(m_c, d_c, ml, mr, dl, dr, rhy);
  Vec k;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const T fx_hi = face_flux<T>(d_c[e], d_up[e], m_c[e], m_up[e], rhx);
    const T fx_lo = face_flux<T>(d_dn[e], d_c[e], m_dn[e], m_c[e], rhx);
    k[e] = div_sum<T>(fx_hi - fx_lo, rhx, dy[e]);
  }
  return k;
}

template <typename T, int TX>
constexpr size_t fused_lds_bytes() {
  constexpr int V = VecOf<T>::V;
  constexpr int HV = 4 / V;            // halo vectors per side (4 columns)
  constexpr int PV = kLanesPerRow + 2 * HV;
  return ((size_t)(TX + 8) * PV * V + (size_t)(TX + 6) * PV * V + 4 * V) * sizeof(T);
}

// NT threads = NT/32 thread rows of kLanesPerRow lanes, RPT tile rows each.  A taller tile re-evaluates
// less (stage A on tile+2, mu on tile+3): 16 rows -> x1.46 mu_A, x1.33 k_A; 32 rows -> x1.26, x1.17.
// Growing RPT pays for that in VGPRs (RPT 4: 132-156, 3 waves/SIMD); growing NT does not.
#ifndef PDEOPT_PAIR_WAVES_ATTR
#define PDEOPT_PAIR_WAVES_ATTR
#endif
// (fp64 PAIR_34 sits one register above the 80 that would let three 512-thread workgroups share a CU; compiled
// for 6 waves per SIMD it takes 78, no scratch -- and measures 504 vs 507 env-steps/s: occupancy is not its limit)
template <typename T, int CL, int PAIR, int RPT, bool RAGGED, int NT>
__global__ __launch_bounds__(NT) PDEOPT_PAIR_WAVES_ATTR void stage_pair_kernel(const PairArgs<T> a, const int tiles_i,
                                                        const int tiles_j, const int nblk,
                                                        const int xcd_remap) {
  using Vec = typename VecOf<T>::type;
  constexpr int V = VecOf<T>::V;
  constexpr int HV = 4 / V;
  constexpr int TX = (NT / kLanesPerRow) * RPT;
  constexpr int PV = kLanesPerRow + 2 * HV;
  constexpr int P = PV * V;
  constexpr int TY = kLanesPerRow * V;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* const sU = reinterpret_cast<T*>(smem_raw) + V;   // rows: tile row + 4, cols: tile col + HV*V
  T* const sMu = sU + (TX + 8) * P + V;               // rows: tile row + 3

  int ti, tj, b;
  decode_tile(blockIdx.x, tiles_i, tiles_j, nblk, xcd_remap, &ti, &tj, &b);
  if (tile_skipped(a.part, ti, tj, tiles_i, tiles_j)) return;
  const int i0 = ti * TX;
  const int j0 = tj * TY;

  const Geo& g = a.g;
  const int64_t ld = g.ld;
  const int64_t base = (int64_t)b * g.bstride + g.off;
  const EnvParams<T>& p = a.ep[b];
  const T* __restrict__ in = a.in + base;
  const T kap = p.kappa;

  const int tid = threadIdx.x;
  const int lx = tid & 31;
  const int ly = tid >> 5;
  const int r0 = ly * RPT;
  const int cvo = lx + HV;  // this thread's vector column in the LDS arrays

  // ring vector of this thread (tile + 2 ring minus the tile): 4 full rows + 2 side vectors per row
  constexpr int kRingRowVecs = kLanesPerRow + 2;
  constexpr int kRingTop = 4 * kRingRowVecs;
  constexpr int kRing = kRingTop + 2 * TX;
  static_assert(kRing <= NT, "ring must fit one pass");
  int ring_r = 0, ring_cv = 0;  // tile row / LDS vector column
  const bool has_ring = tid < kRing;
  if (tid < kRingTop) {
    const int q = tid / kRingRowVecs;
    ring_r = (q < 2) ? (q - 2) : (TX + q - 2);
    ring_cv = HV - 1 + (tid - q * kRingRowVecs);
  } else if (has_ring) {
    const int t2 = tid - kRingTop;
    ring_r = t2 >> 1;
    ring_cv = (t2 & 1) ? (HV + kLanesPerRow) : (HV - 1);
  }

  constexpr bool ragged = RAGGED;  // compile-time: the divisible case pays nothing for the masks
  auto wrap_row = [&](int gi) { return g.periodic ? tile_wrap(gi, g.nx, ragged) : gi; };
  auto wrap_col = [&](int gj) { return g.periodic ? tile_wrap(gj, g.ny, ragged) : gj; };
  // ragged tiles: every lane computes, only cells inside the grid are loaded pointwise / stored
  const bool col_ok = !RAGGED || (j0 + lx * V) < g.ny;
  auto cell_ok = [&](int r) { return !RAGGED || (col_ok && (i0 + r0 + r) < g.nx); };

  // ---- prefetch pointwise operands (PAIR_34: y on own cells + ring, acc on own cells)
  const int64_t pidx0 = base + (int64_t)(i0 + r0) * ld + (j0 + lx * V);
  Vec ybase[RPT], accp[RPT], yring;
  if constexpr (PAIR == PAIR_34) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      accp[r] = Vec{};
      if constexpr (RAGGED) {
        // cells of the tile that lie beyond the grid are periodic images: stage B of the cells next
        // to the grid edge reads w there, so their base value is needed (wrapped), not masked
        const int gi = wrap_row(i0 + r0 + r), gj = wrap_col(j0 + lx * V);
        ybase[r] = *reinterpret_cast<const Vec*>(a.y + base + (int64_t)gi * ld + gj);
        if (cell_ok(r)) accp[r] = *reinterpret_cast<const Vec*>(a.acc + pidx0 + r * ld);
      } else {
        ybase[r] = *reinterpret_cast<const Vec*>(a.y + pidx0 + r * ld);
        accp[r] = *reinterpret_cast<const Vec*>(a.acc + pidx0 + r * ld);
      }
    }
    if (has_ring) {
      const int gi = wrap_row(i0 + ring_r);
      const int gj = wrap_col(j0 + (ring_cv - HV) * V);
      yring = *reinterpret_cast<const Vec*>(a.y + base + (int64_t)gi * ld + gj);
    }
  }

  // ---- P1: stage-A input, tile + 4  (PAIR_K: the only stage's input, tile + 2 = LDS rows 2 .. TX+5)
  constexpr int kRow0 = PAIR == PAIR_K ? 2 : 0;
#ifndef PDEOPT_P1_FLAT
  // One tile row per wave and trip: lane l < PV loads vector l of the row.  The row index is wave-uniform, so
  // its periodic wrap and the row offset are scalar work and a trip costs the VALU nothing but the load and the
  // LDS store; the column wrap and the column offset are formed once per thread.  (The flat "vector idx = tid +
  // it NT" mapping needed a division by PV, two wraps and a 64-bit address per vector: ~100 of the ~700 VALU
  // instructions a thread executes per tile.)  A wave reads 34 (36) contiguous vectors of one row.
  bool halo_from_strips = false;  // wave-uniform
  if constexpr (PAIR == PAIR_12 && !RAGGED)
    halo_from_strips = a.nbase[0] != nullptr && (ti == 0 || ti == tiles_i - 1 || tj == 0 || tj == tiles_j - 1);
  if (halo_from_strips) {
    // Fused unpack (decomposed field, edge tiles of the extended launch).  True coordinates (tile interior = [0, nx)
    // x [0, ny)) of a loaded vector: gi = i0 - 4 + row - ext, gj = j0 - HV V + lane V - ext.  Interior cells come
    // from the field, cells of the 8-wide halo from the strip piece of the neighbour they belong to (halo.hip:
    // my halo piece q <- piece FROM[q] of neighbour q), cells beyond the halo (the over-run of the last tile row /
    // column) from the layout's margin -- computed on, never read back.  A vector never straddles two sources (all
    // boundaries are multiples of V).  Per thread: one source pointer + pitch for each row class, chosen by its
    // column class; per trip: the wave-uniform row picks the class.
    constexpr int NW = NT / 64;
    constexpr int H = 8;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if (lane < PV) {
      const int nx = g.nx, ny = g.ny, ext = a.ext;
      const StripOffsets<H> so(nx, ny);
      const T* const mem = in + (int64_t)ext * ld + ext;  // cell (0, 0) of the tile interior
      const int64_t eoff = (int64_t)b * a.strip_env;  // this environment inside a rank's strip
      const int gj = j0 - HV * V + lane * V - ext;
      const T *pT, *pM, *pB;
      int64_t qT, qM, qB;  // pitches
      if (gj >= 0 && gj < ny) {
        pT = a.nbase[0] + eoff + so.bottom + gj; qT = ny;   // up's bottom rows
        pM = mem + gj; qM = ld;
        pB = a.nbase[1] + eoff + so.top + gj; qB = ny;      // down's top rows
      } else if (gj < 0) {
        const int c = gj + H;
        pT = a.nbase[4] + eoff + so.br + c; pM = a.nbase[2] + eoff + so.right + c; pB = a.nbase[6] + eoff + so.tr + c;
        qT = qM = qB = H;
      } else if (gj < ny + H) {
        const int c = gj - ny;
        pT = a.nbase[5] + eoff + so.bl + c; pM = a.nbase[3] + eoff + so.left + c; pB = a.nbase[7] + eoff + so.tl + c;
        qT = qM = qB = H;
      } else {  // beyond the halo: the margin
        pT = mem + gj - (int64_t)H * ld; pM = mem + gj; pB = mem + gj + (int64_t)nx * ld;
        qT = qM = qB = ld;
      }
      const T* const pO = mem + gj;
      T* const lds = sU + lane * V;
      constexpr int kRows = TX + 8, kTrips = (kRows + NW - 1) / NW;
      Vec f[kTrips];
#pragma unroll
      for (int k = 0; k < kTrips; ++k) {
        int row = wave + k * NW;
        if constexpr (kRows % NW != 0) row = row < kRows ? row : kRows - 1;
        const int gi = i0 - 4 + row - ext;  // wave-uniform
        const T* src;
        if (gi < 0) src = pT + (int64_t)(gi + H) * qT;
        else if (gi < nx) src = pM + (int64_t)gi * qM;
        else if (gi < nx + H) src = pB + (int64_t)(gi - nx) * qB;
        else src = pO + (int64_t)gi * ld;
        f[k] = *reinterpret_cast<const Vec*>(src);
      }
#pragma unroll
      for (int k = 0; k < kTrips; ++k) {
        const int row = wave + k * NW;
        if (kRows % NW == 0 || k + 1 < kTrips || row < kRows) *reinterpret_cast<Vec*>(lds + row * P) = f[k];
      }
    }
  } else {
    constexpr int NW = NT / 64;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if (lane < PV) {
      const int gj = wrap_col(j0 - HV * V + lane * V);
      const T* __restrict__ colp = in + gj;
      T* const lds = sU + lane * V;
      // all loads of the thread first, then the LDS stores: the trips' latencies overlap.  Every trip loads
      // (a wave past the last row re-reads that row: unconditional loads keep the registers simple), only
      // the store of a partial last trip is guarded -- by a wave-uniform condition
      constexpr int kRows = TX + 8 - 2 * kRow0, kTrips = (kRows + NW - 1) / NW;
      Vec f[kTrips];
#pragma unroll
      for (int k = 0; k < kTrips; ++k) {
        int row = kRow0 + wave + k * NW;
        if constexpr (kRows % NW != 0) row = row < kRow0 + kRows ? row : kRow0 + kRows - 1;
        const int gi = wrap_row(i0 - 4 + row);
        if (PDEOPT_ABL(a, 8)) {
          for (int e = 0; e < V; ++e) f[k][e] = T(0.5) + T(1e-4) * T(gj + e);
        } else {
          f[k] = *reinterpret_cast<const Vec*>(colp + (int64_t)gi * ld);
        }
      }
#pragma unroll
      for (int k = 0; k < kTrips; ++k) {
        const int row = kRow0 + wave + k * NW;
        if (kRows % NW == 0 || k + 1 < kTrips || row < kRow0 + kRows) *reinterpret_cast<Vec*>(lds + row * P) = f[k];
      }
    }
  }
#else
  constexpr int kLoadVecs = (TX + 8 - 2 * kRow0) * PV;
#pragma unroll
  for (int it = 0; it < (kLoadVecs + NT - 1) / NT; ++it) {
    const int idx = tid + it * NT;
    if (idx < kLoadVecs) {
      const int row = idx / PV + kRow0;
      const int cv = idx - (row - kRow0) * PV;
      const int gi = wrap_row(i0 - 4 + row);
      const int gj = wrap_col(j0 - HV * V + cv * V);
      if (PDEOPT_ABL(a, 8)) {
        Vec f;
        for (int e = 0; e < V; ++e) f[e] = T(0.5) + T(1e-4) * T(gj + e);
        *reinterpret_cast<Vec*>(sU + row * P + cv * V) = f;
      } else {
        *reinterpret_cast<Vec*>(sU + row * P + cv * V) = *reinterpret_cast<const Vec*>(in + (int64_t)gi * ld + gj);
      }
    }
  }
#endif
  __syncthreads();

  // mu on `nrows` rows starting at mu-row `rm0` (mu row rm <-> sU row rm + 1)
  // Linear-logit class (the headline's closures; the cubic classes gain less, +3 %, and keep the literal
  // form): mu = mu_h(c) - kappa lap c is evaluated as
  //   fA (c_x+ + c_x-) + fB (c_y+ + c_y-) + (q1 c + q0 + logit c),  fA = -kappa/hx^2, fB = -kappa/hy^2,
  //   q1 = coef1 - 2 (fA + fB)
  // -- the same expression re-associated with the constants folded per environment: 3 VALU instructions
  // less per evaluation (+4 % same-box on the headline once the build stopped packing fp32 pairs; it had
  // measured -3 % on the packed build).  Rounding differs from the literal form by a few ulp of the state.
#ifndef PDEOPT_PAIR_FOLD_MU
#define PDEOPT_PAIR_FOLD_MU 1
#endif
  constexpr bool FOLD_MU = PDEOPT_PAIR_FOLD_MU && CL == CL_LOGIT1;
  T fA = T(0), fB = T(0), q1 = T(0);
  if constexpr (FOLD_MU) {
    fA = -kap * a.rhx2;
    fB = -kap * a.rhy2;
    q1 = p.mu[1] - T(2) * (fA + fB);
  }
  auto mu_pass = [&](const int rm0, const int nrows) {
    const int nvec = nrows * PV;
    const int lane = tid & 63;
    // all (<= 3 at fp32) trips unrolled: independent closure evaluations interleave and hide the
    // v_rcp / v_log latencies; +1.7 % same-box against one trip at a time, same VGPR count
#ifndef PDEOPT_MU_UNROLL
#define PDEOPT_MU_UNROLL 3
#endif
#pragma unroll PDEOPT_MU_UNROLL
    for (int base0 = 0; base0 < nvec; base0 += NT) {
      const int idx_raw = base0 + tid;
#ifdef PDEOPT_DPP_EXCHANGE
      // uniform trip count (every lane takes part in the DPP shifts); lanes past the end recompute
      // the last vector and do not store
      const int idx = idx_raw < nvec ? idx_raw : nvec - 1;
#else
      // a wave that lies entirely past the end leaves (mu passes are half of this kernel's VALU work
      // and the last trip is only 40-90 % populated); a partly populated wave runs under its exec mask
      if (idx_raw >= nvec) break;
      const int idx = idx_raw;
#endif
      const int rr = idx / PV;
      const int cv = idx - rr * PV;
      const int rm = rm0 + rr;
      const T* c_ = sU + (rm + 1) * P + cv * V;
      const Vec c = *reinterpret_cast<const Vec*>(c_);
      const Vec xp = *reinterpret_cast<const Vec*>(c_ + P);
      const Vec xm = *reinterpret_cast<const Vec*>(c_ - P);
      // consecutive lanes hold consecutive vectors of the (unpadded) row-major array: the left /
      // right scalar neighbours sit in the adjacent lanes' registers; only the wave's end lanes read LDS
#ifdef PDEOPT_DPP_EXCHANGE  // measured 6 % SLOWER than the conflicted LDS reads (VALU-bound kernel, +6 VGPRs)
      T left = lane_from_prev(T(0), c[V - 1]);
      T right = lane_from_next(T(0), c[0]);
      if (lane == 0) left = c_[-1];
      if (lane == 63) right = c_[V];
#else
      const T left = nb_left<T, Vec, V>(c_), right = nb_right<T, Vec, V>(c_);
      (void)lane;
#endif
      Vec m;
      if (PDEOPT_ABL(a, 1)) {
        m = c + xp + xm + left + right;
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T ym = (e == 0) ? left : c[e - 1];
          const T yp = (e == V - 1) ? right : c[e + 1];
          if constexpr (FOLD_MU)
            m[e] = fA * (xp[e] + xm[e]) + (fB * (yp + ym) + (q1 * c[e] + p.mu[0] + t_logit<T>(c[e])));
          else
            m[e] = eval_mu<T, CL>(a.mu, p.mu, c[e]) - kap * lap_at<T>(c[e], xp[e], xm[e], yp, ym, a.rhx2, a.rhy2);
        }
      }
      if (idx_raw < nvec) *reinterpret_cast<Vec*>(sMu + rm * P + cv * V) = m;
    }
  };

  // k at one vector: tile row r, LDS vector column cv
  auto k_at = [&](const int r, const int cv, Vec* centre) -> Vec {
    const T* mp = sMu + (r + 3) * P + cv * V;
    const T* up = sU + (r + 4) * P + cv * V;
    const Vec u_c = *reinterpret_cast<const Vec*>(up);
    if (centre) *centre = u_c;
    return flux_divergence<T, CL, Vec, V>(
        a.mob, p.mob, *reinterpret_cast<const Vec*>(mp - P), *reinterpret_cast<const Vec*>(mp),
        *reinterpret_cast<const Vec*>(mp + P), *reinterpret_cast<const Vec*>(up - P), u_c,
        *reinterpret_cast<const Vec*>(up + P), nb_left<T, Vec, V>(mp), nb_right<T, Vec, V>(mp), nb_left<T, Vec, V>(up),
        nb_right<T, Vec, V>(up), a.rhx, a.rhy);
  };

  // k on the own micro-tile (RPT rows x 1 vector), marching down the rows so every mu / u row is
  // read once and every x-face flux is formed once
  auto march = [&](Vec* kout, Vec* centre) {
    const T* mp = sMu + (r0 + 2) * P + cvo * V;  // row r0 - 1
    const T* up = sU + (r0 + 3) * P + cvo * V;
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
      // left / right neighbours of the centre row: adjacent lanes' registers (mu and the mobility
      // they already evaluated); the first / last lane of a tile row takes the halo from LDS
#ifdef PDEOPT_DPP_EXCHANGE  // measured 6 % SLOWER than the conflicted LDS reads (VALU-bound kernel, +6 VGPRs)
      T ml = lane_from_prev(T(0), m_c[V - 1]), mr = lane_from_next(T(0), m_c[0]);
      T dl = lane_from_prev(T(0), d_c[V - 1]), dr = lane_from_next(T(0), d_c[0]);
      if (lx == 0) {
        ml = mp[-1];
        dl = eval_mob<T, CL>(a.mob, p.mob, up[-1]);
      }
      if (lx == kLanesPerRow - 1) {
        mr = mp[V];
        dr = eval_mob<T, CL>(a.mob, p.mob, up[V]);
      }
#else
      const T ml = nb_left<T, Vec, V>(mp), mr = nb_right<T, Vec, V>(mp);
      const T dl = eval_mob<T, CL>(a.mob, p.mob, nb_left<T, Vec, V>(up)), dr = eval_mob<T, CL>(a.mob, p.mob, nb_right<T, Vec, V>(up));
#endif
      const Vec dy = div_y<T, Vec, V>(m_c, d_c, ml, mr, dl, dr, a.rhy);
      Vec fx_hi, k;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        fx_hi[e] = face_flux<T>(d_c[e], d_hi[e], m_c[e], m_hi[e], a.rhx);
        k[e] = div_sum<T>(fx_hi[e] - fx_lo[e], a.rhx, dy[e]);  // same contraction as flux_divergence()
      }
      kout[r] = k;
      if (centre) centre[r] = u_c;
      m_c = m_hi;
      u_c = u_hi;
      d_c = d_hi;
      fx_lo = fx_hi;
      mp += P;
      up += P;
    }
  };

  Vec yown[RPT];  // PAIR_12: y on the own cells (stage A's centre values), used again by the stores
  if constexpr (PAIR != PAIR_K) {
  // ---- P2: mu_A on tile + 3 (mu rows 0 .. TX+5)
  mu_pass(0, TX + 6);
  __syncthreads();

  // ---- P3: k_A on one ring vector, then on the own micro-tile.  The ring goes FIRST: its non-marching
  // flux_divergence needs ~60 transient registers, and after the march w_own / yown / accp are live --
  // in the other order this phase is the VGPR peak of the kernel (v95 / v115).
  Vec w_own[RPT], w_ring;
  if (has_ring) {
    Vec uc;
    Vec kA;
    if (PDEOPT_ABL(a, 4))
      kA = uc = *reinterpret_cast<const Vec*>(sU + (ring_r + 4) * P + ring_cv * V);
    else
      kA = k_at(ring_r, ring_cv, &uc);
    if constexpr (PAIR == PAIR_12)
      w_ring = uc + a.aA * kA;
    else
      w_ring = yring + a.aA * kA;
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    Vec kA[RPT];
    if (PDEOPT_ABL(a, 2)) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) kA[r] = yown[r] = *reinterpret_cast<const Vec*>(sU + (r0 + r + 4) * P + cvo * V);
    } else {
      march(kA, yown);  // PAIR_12: the stage-A input IS y
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if constexpr (PAIR == PAIR_12) {
        w_own[r] = yown[r] + a.aA * kA[r];
        accp[r] = yown[r] + a.bA * kA[r];
      } else {
        w_own[r] = ybase[r] + a.aA * kA[r];
        accp[r] = accp[r] + a.bA * kA[r];
      }
    }
  }
  __syncthreads();

  // ---- P4: w -> sU in place (tile + 2)
#pragma unroll
  for (int r = 0; r < RPT; ++r) *reinterpret_cast<Vec*>(sU + (r0 + r + 4) * P + cvo * V) = w_own[r];
  if (has_ring) *reinterpret_cast<Vec*>(sU + (ring_r + 4) * P + ring_cv * V) = w_ring;
  __syncthreads();

  }  // PAIR != PAIR_K

  // ---- P5: mu_B on tile + 1 (mu rows 2 .. TX+3)
  mu_pass(2, TX + 2);
  __syncthreads();

  // ---- P6: k_B, stage updates, stores
  Vec kB[RPT];
  if (PDEOPT_ABL(a, 2)) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) kB[r] = *reinterpret_cast<const Vec*>(sMu + (r0 + r + 3) * P + cvo * V);
  } else {
    march(kB, nullptr);
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    if (!cell_ok(r)) continue;
    if (PDEOPT_ABL(a, 16) && kB[r][0] != T(12345.678)) continue;
    const int64_t idx = pidx0 + r * ld;
    if constexpr (PAIR == PAIR_K) {
      *reinterpret_cast<Vec*>(a.out + idx) = kB[r];
    } else if constexpr (PAIR == PAIR_12) {
      *reinterpret_cast<Vec*>(a.out + idx) = yown[r] + a.aB * kB[r];
      *reinterpret_cast<Vec*>(a.acc_out + idx) = accp[r] + a.bB * kB[r];
      if (halo_from_strips) {
        // the unpack's other half: own cells outside the tile interior go back into the field's halo frame, where
        // the second stage pair reads y (pointwise, tile + 2)
        const int gi = i0 + r0 + r - a.ext, gj = j0 + lx * V - a.ext;
        if (gi < 0 || gi >= g.nx || gj < 0 || gj >= g.ny) *reinterpret_cast<Vec*>(const_cast<T*>(a.in) + idx) = yown[r];
      }
    } else {
      const Vec ynew = accp[r] + a.bB * kB[r];
      *reinterpret_cast<Vec*>(a.out + idx) = ynew;
      // fused pack (decomposed field): wave-uniform test first, interior tiles skip everything
      if (a.strip != nullptr && (ti == 0 || ti == tiles_i - 1 || tj == 0 || tj == tiles_j - 1)) {
        constexpr int H = 8;
        static_assert(H % V == 0, "a vector never straddles two strip pieces");
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

// --------------------------------------------------------------------------------------------- host

template <typename T>
bool fused_supported(const pdeopt_ctx* ctx) {
  if (ctx->opt_fuse_stages < 0) return false;
  if (ctx->prob.equation != PDEOPT_EQ_CAHN_HILLIARD && ctx->prob.equation != PDEOPT_EQ_ALLEN_CAHN)
    return false;
  if (!tiled_supported<T>(ctx)) return false;
  return classify_closures(ctx->prob.mu, ctx->prob.mob) != CL_GENERIC;
}

template <typename T, int CL, int PAIR, int RPT>
int launch_pair_ac_inst(pdeopt_ctx* ctx, const PairArgs<T>& s);  // stencil_fused_ac.hpp

template <typename T, int CL, int PAIR, int RPT>
int launch_pair_ch_inst(pdeopt_ctx* ctx, const PairArgs<T>& s, int ext);

template <typename T, int CL, int PAIR, int RPT>
int launch_pair_inst(pdeopt_ctx* ctx, const PairArgs<T>& s, int ext = 0) {
  if (ctx->prob.equation == PDEOPT_EQ_ALLEN_CAHN) return launch_pair_ac_inst<T, CL, PAIR, RPT>(ctx, s);
  return launch_pair_ch_inst<T, CL, PAIR, RPT>(ctx, s, ext);
}

template <typename T, int CL, int PAIR, int RPT>
int launch_pair_ch_inst(pdeopt_ctx* ctx, const PairArgs<T>& s, int ext) {
  // CH: 2 rows per thread always; 32-row tiles (RPT == 4 on this dispatch axis) are 512-thread blocks
  constexpr int V = VecOf<T>::V;
  constexpr int NT = RPT == 4 ? 512 : 256;
  constexpr int TX = (NT / kLanesPerRow) * 2;
  const pdeopt_problem& p = ctx->prob;
  // ext > 0 (halo-8 layout, PAIR_12): the launch covers the tile + ext ring in WHOLE workgroup tiles starting at
  // cell (-ext, -ext) -- the caller shifted the pointers; the last tile row / column over-runs into the layout's
  // tail margin (common.hpp: pad_rows / pad_ld), nothing is masked
  const int tiles_i = (p.nx + 2 * ext + TX - 1) / TX;
  const int tiles_j = (p.ny + 2 * ext + kLanesPerRow * V - 1) / (kLanesPerRow * V);
  const int64_t nblk64 = (int64_t)tiles_i * tiles_j * ctx->win_n;
  if (nblk64 > 0x7fffffffLL) return fail(ctx, PDEOPT_EINVAL, "too many tiles");
  const int nblk = (int)nblk64;
  const size_t lds = fused_lds_bytes<T, TX>();
  const bool ragged = ext == 0 && (p.nx % TX != 0 || p.ny % (kLanesPerRow * V) != 0);
  if (ragged)
    hipLaunchKernelGGL((stage_pair_kernel<T, CL, PAIR, 2, true, NT>), dim3(nblk), dim3(NT), lds, ctx->stream, s, tiles_i,
                       tiles_j, nblk, tile_flags(nblk, tiles_i, tiles_j));
  else
    hipLaunchKernelGGL((stage_pair_kernel<T, CL, PAIR, 2, false, NT>), dim3(nblk), dim3(NT), lds, ctx->stream, s, tiles_i,
                       tiles_j, nblk, tile_flags(nblk, tiles_i, tiles_j));
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

}  // namespace pdeopt
