// Adaptive Tsit5 with the PID step-size controller INSIDE the kernel, for grids that fit one compute unit's LDS
// (diffrax.Tsit5 + diffrax.PIDController as the reference's tests use them:
// tests/test_solvers.py:81,263).
//
// The host-driven loop (pde_opt_amd/integrate.py: _solve_adaptive) pays, per trial step, 8 stage launches, an error
// reduction and a device->host read before it can choose the next step size: 100-127 us per step on a 64^2 grid whose
// seven stages compute for under 10.  Here one workgroup owns one environment for the whole t0 -> t1 solve: the state
// and the seven slopes of the thread's vectors stay in registers, the stage input (and mu for Cahn-Hilliard) in LDS
// (stencil_small.hpp: SmallTile), the scaled error norm is a block reduction, and every thread runs the same
// controller arithmetic on the same norm -- no launch, no host round trip between steps.  Dense output at the
// requested save times is written from the accepted step's slopes as the solve passes them.
//
// Each environment runs its own controller (its own t, dt, accept / reject), which is what batch == 1 and
// PIDController(per_environment=True) ask for; a step size shared by several environments needs a reduction across
// workgroups and stays on the host-driven path.
#pragma once
#include "stencil_small.hpp"

namespace pdeopt {

// Tsit5: Ch. Tsitouras, Comput. Math. Appl. 62 (2011) 770-775 -- the published coefficients diffrax ships.
constexpr double kTsA[6][6] = {
    {0.161, 0, 0, 0, 0, 0},
    {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
// abscissae of stages 2..7 (row sums of kTsA)
constexpr double kTsC[6] = {0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
constexpr double kTsE[7] = {0.00178001105222577714, 0.0008164344596567469, -0.007880878010261995, 0.1447110071732629,
                            -0.5823571654525552,    0.45808210592918697,   -1.0 / 66.0};

// b_i(theta) of the 4th-order continuous extension (Tsitouras 2011, section 4); b_i(1) = the 5th-order weights
__host__ __device__ inline void tsit5_dense_weights(double th, double* b) {
  const double t2 = th * th;
  b[0] = -1.0530884977290216 * th * (th - 1.3299890189751412) * (t2 - 1.4364028541716351 * th + 0.7139816917074209);
  b[1] = 0.1017 * t2 * (t2 - 2.1966568338249754 * th + 1.2949852507374631);
  b[2] = 2.490627285651252793 * t2 * (t2 - 2.38535645472061657 * th + 1.57803468208092486);
  b[3] = -16.54810288924490272 * (th - 1.21712927295533244) * (th - 0.61620406037800089) * t2;
  b[4] = 47.37952196281928122 * (th - 1.203071208372362603) * (th - 0.658047292653547382) * t2;
  b[5] = -34.87065786149660974 * (th - 1.2) * (th - 0.666666666666666667) * t2;
  b[6] = 2.5 * (th - 1.0) * (th - 0.6) * t2;
}

// the controller's constants, folded on the host (integrate.py: _pid_update)
struct PidConsts {
  double rtol, atol;
  double k1, k2, k3;  // exponents of 1/err, 1/err_prev, 1/err_prev_prev
  double factormin, factormax, safety;
  double dtmin, dtmax;  // -inf / +inf: none
};

template <typename T>
struct SmallTsit5Args {
  T* y;  // state of the first environment, dense (nx, ny) per environment
  int nx, ny;
  int64_t bstride;
  T rhx, rhy, rhx2, rhy2;
  const EnvParams<T>* ep;
  ClosureSpec mu, mob;
  double t0, t1, dt0;
  PidConsts pid;
  int64_t max_steps;
  int n_save;             // save times > t0, ascending
  const double* save_ts;  // device
  T* save_out;            // device [n_save][batch][nx * ny]; slots the solve never reaches keep their NaN fill
  int64_t save_stride;    // elements between two save points
  pdeopt_tsit5_stats* stats;  // device [batch]
  int red_off;            // byte offset of the reduction scratch in LDS
};

#ifndef PDEOPT_PID_POW
#define PDEOPT_PID_POW(b, e) exp((e) * log(b))
#endif
// diffrax.PIDController's factor for one scaled error norm: the same branches as integrate.py: _pid_update
__device__ inline double pid_term(double base, double expo, const PidConsts& c) {
  if (expo == 0.0) return 1.0;
  // base^expo as exp(expo log base): |expo log base| < 1 for any factor the clipping below lets through, so the result
  // carries ~1e-16 relative error like pow's, at a third of its instructions (one lane runs this while every other wave
  // of the solve waits: stencil_coop_adaptive.hpp)
  if (base > 0 && base < __builtin_inf()) return PDEOPT_PID_POW(base, expo);
  return base > 0 ? c.factormax : c.factormin;
}

// LDS: [stage input A][mu (Cahn-Hilliard)][stage input B][state y, when Y_LDS][reduction scratch].  The stage input is
// double-buffered: a stage's update writes the next input straight into the other buffer while neighbours may still
// be reading the current one -- one barrier per stage instead of two, and no register copy of the input (the
// candidate y1 is read back from LDS for the error norm and the accept).  Y_LDS (4 vectors per thread): the state
// lives in LDS as well, each thread reading and writing only its own vectors -- 16 registers less, without which
// the 4-vector form spilled.
template <typename T, int EQ, int CL, int KMAX, int NTMAX>
__global__ __launch_bounds__(NTMAX) void small_tsit5_kernel(const SmallTsit5Args<T> a) {
  using Tile = SmallTile<T, EQ, CL, KMAX>;
  using Vec = typename Tile::Vec;
  constexpr int V = Tile::V;
  constexpr bool Y_LDS = KMAX >= 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* const red = reinterpret_cast<double*>(smem_raw + a.red_off);
  const int b = blockIdx.x;
  T* const yg = a.y + (int64_t)b * a.bstride;
  Tile tile;
  Vec y[Y_LDS ? 1 : KMAX], k[6][KMAX];  // k7 lives in the idle input buffer (below)
  {
    Vec y0[KMAX];
    tile.init(smem_raw, a.nx, a.ny, a.ep + b, a.mu, a.mob, a.rhx, a.rhy, a.rhx2, a.rhy2, yg, y0);
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if constexpr (!Y_LDS) y[j] = y0[j];
  }
  T* sNext = tile.sU + 2 * tile.cells;  // past mu (Allen-Cahn leaves that array unused: the layout is one for both)
  T* const sY = tile.sU + 3 * tile.cells;
  if constexpr (Y_LDS) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) *reinterpret_cast<Vec*>(sY + tile.oc[j]) = *reinterpret_cast<const Vec*>(tile.sU + tile.oc[j]);
  }
  auto state = [&](int j) -> Vec {
    if constexpr (Y_LDS) return *reinterpret_cast<const Vec*>(sY + tile.oc[j]);
    else return y[j];
  };
  // the next stage input is complete in sNext once every thread is here; the current one is then free to be overwritten
  auto flip = [&]() {
    __syncthreads();
    T* const tmp = tile.sU;
    tile.sU = sNext;
    sNext = tmp;
  };
  const T rtol = T(a.pid.rtol), atol = T(a.pid.atol);
  const int nwaves = (blockDim.x + 63) >> 6;
  const double inv_cells = 1.0 / (double)tile.cells;

  tile.rhs([&](int j, const Vec kv) { k[0][j] = kv; });  // k1 = f(y0); afterwards k1 is the previous step's k7 (FSAL)

  double t = a.t0, dt = a.dt0, prev_inv = 1.0, prev_prev_inv = 1.0;
  int64_t accepted = 0, rejected = 0;
  int qi = 0, status = PDEOPT_TSIT5_DONE;
  double next_tq = a.n_save > 0 ? a.save_ts[0] : __builtin_inf();  // a.save_ts[qi]: re-read when qi moves, not per step
  while (t < a.t1) {
    if (accepted + rejected >= a.max_steps) {
      status = PDEOPT_TSIT5_MAX_STEPS;
      break;
    }
    const double h = fmin(dt, a.t1 - t);
    if (!(h > 0.0) || t + h == t) {  // a step that cannot advance t (dt underflow, NaN): the loop would never end
      status = PDEOPT_TSIT5_STALLED;
      break;
    }
    // stage 2 input from k1 alone; stages 2..6 compute k_s and the next input y + h sum_j a_{s+1,j} k_j while k_s is
    // in registers (the association of lincomb_kernel / launch_stage_lc: r = y; r += c_j k_j in order); the input of
    // stage 7 is the 5th-order candidate y1.  (The k1 = f(y0) pass above, a rejected step's and an accepted step's
    // tail all end in a barrier-free stretch after the last read of the current input: writing sNext is safe.)
    {
      const T c0 = T(h * kTsA[0][0]);
#pragma unroll
      for (int j = 0; j < KMAX; ++j) *reinterpret_cast<Vec*>(sNext + tile.oc[j]) = state(j) + c0 * k[0][j];
    }
    flip();
#pragma unroll
    for (int s = 1; s <= 5; ++s) {
      T c[6];
#pragma unroll
      for (int i = 0; i <= s; ++i) c[i] = T(h * kTsA[s][i]);
      tile.rhs([&](int j, const Vec kv) {
        k[s][j] = kv;
        Vec r = state(j);
#pragma unroll
        for (int i = 0; i <= s; ++i) r = r + c[i] * k[i][j];
        *reinterpret_cast<Vec*>(sNext + tile.oc[j]) = r;
      });
      flip();
    }
    // stage 7: k7 = f(y1), and with it the scaled error: sum ((h sum_j e_j k_j) / (atol + rtol max(|y|, |y1|)))^2
    double part = 0.0;
    {
      T ce[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) ce[i] = T(h * kTsE[i]);
      // k7 goes to the thread's own slots of the idle input buffer: nobody else touches sNext until the next flip
      tile.rhs([&](int j, const Vec kv) {
        *reinterpret_cast<Vec*>(sNext + tile.oc[j]) = kv;
        Vec e = Vec{};
#pragma unroll
        for (int i = 0; i < 6; ++i) e = e + ce[i] * k[i][j];
        e = e + ce[6] * kv;
        const Vec y0 = state(j), y1 = *reinterpret_cast<const Vec*>(tile.sU + tile.oc[j]);
        double sq = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const T sc = atol + rtol * fmax(fabs(y0[v]), fabs(y1[v]));
          const double q = (double)(e[v] / sc);
          sq += q * q;
        }
        if (tile.own(j)) part += sq;
      });
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_down(part, sft, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();  // also: every read of stage 7's input by a neighbour is done
    double sum = 0.0;
    for (int i = 0; i < nwaves; ++i) sum += red[i];  // the same order in every thread: one norm for the workgroup
    const double err = sqrt(sum * inv_cells);        // diffrax rms_norm

    const bool keep = err < 1.0;  // a NaN norm rejects
    const double inv = (err > 0.0 && err < __builtin_inf()) ? 1.0 / err : (err == 0.0 ? __builtin_inf() : 0.0);
    double f = pid_term(inv, a.pid.k1, a.pid) * pid_term(prev_inv, a.pid.k2, a.pid) * pid_term(prev_prev_inv, a.pid.k3, a.pid);
    f = fmin(a.pid.factormax, fmax(a.pid.factormin, a.pid.safety * f));
    if (!keep) f = fmin(1.0, f);
    if (keep) {
      ++accepted;
      const double t_new = t + h;
      // dense output from the accepted step's seven slopes (4th-order interpolant), before they are recycled
      while (qi < a.n_save) {
        const double tq = next_tq;
        if (!(tq <= t_new + 1e-14 * fmax(1.0, fabs(t_new)))) break;
        double bw[7];
        tsit5_dense_weights(fmin(1.0, fmax(0.0, (tq - t) / h)), bw);
        T* const out = a.save_out + (int64_t)qi * a.save_stride + (int64_t)b * tile.cells;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          Vec r = state(j);
#pragma unroll
          for (int i = 0; i < 6; ++i) r = r + T(h * bw[i]) * k[i][j];
          r = r + T(h * bw[6]) * *reinterpret_cast<const Vec*>(sNext + tile.oc[j]);
          if (tile.own(j)) *reinterpret_cast<Vec*>(out + tile.oc[j]) = r;
        }
        ++qi;
        next_tq = qi < a.n_save ? a.save_ts[qi] : __builtin_inf();
      }
      // (state in LDS: a thread that redoes the last vector -- every thread, when the vector count is 3 x 512 -- must not
      // overwrite that vector's y with y1 while another wave still reads y for the dense output above: round 4, found
      // by the 64 x 96 shape as wrong SAVED values in the last four cells)
      if constexpr (Y_LDS) __syncthreads();
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        const Vec y1 = *reinterpret_cast<const Vec*>(tile.sU + tile.oc[j]);
        if constexpr (Y_LDS) *reinterpret_cast<Vec*>(sY + tile.oc[j]) = y1;
        else y[j] = y1;
        k[0][j] = *reinterpret_cast<const Vec*>(sNext + tile.oc[j]);  // FSAL
      }
      // A thread without a k-th vector of its own redoes the LAST vector (SmallTile::init) and, from two vectors per
      // thread on, sits in another wave than that vector's owner: the next iteration's first store into sNext (the
      // stage-2 input, into the slot k7 was just read from) must not overtake the owner's read.  `keep` is the same
      // in every thread, so the barrier is uniform.  (One vector per thread: the duplicates share the owner's wave.)
      if constexpr (KMAX >= 2) __syncthreads();
      t = t_new < a.t1 - 1e-14 * fmax(1.0, fabs(a.t1)) ? t_new : a.t1;
      prev_prev_inv = prev_inv;
      prev_inv = inv;
    } else {
      ++rejected;
    }
    dt = fmin(a.pid.dtmax, fmax(a.pid.dtmin, h * f));
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (tile.own(j)) *reinterpret_cast<Vec*>(yg + tile.oc[j]) = state(j);
  if (threadIdx.x == 0) {
    pdeopt_tsit5_stats st;
    st.t = t;
    st.dt = dt;
    st.accepted = accepted;
    st.rejected = rejected;
    st.status = status;
    st.saved = qi;
    a.stats[b] = st;
  }
}

// --------------------------------------------------------------------------------------------- host

constexpr int kSmallTsit5MaxVec = 4 * 512;  // 9 register-resident fields per vector: 4 vectors per thread of 512

template <typename T>
bool small_tsit5_supported(const pdeopt_ctx* ctx) {
  constexpr int V = VecOf<T>::V;
  if (ctx->opt_small_persist < 0 || ctx->opt_kernel_path == 1 || ctx->opt_debug_ablate) return false;
  if (!small_supported<T>(ctx)) return false;
  const SmallDims d = small_dims(ctx->prob);
  return (int64_t)d.nx * (d.ny / V) <= (int64_t)kSmallTsit5MaxVec;
}

template <typename T, int EQ, int CL>
int launch_small_tsit5_k(pdeopt_ctx* ctx, const SmallTsit5Args<T>& s, int nt, int kmax, size_t lds) {
  auto go = [&](auto kern) -> int {
    if (lds > 48 * 1024)
      PDEOPT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(ctx->prob.batch), dim3(nt), lds, ctx->stream, s);
    PDEOPT_HIP_CHECK(ctx, hipGetLastError());
    return PDEOPT_OK;
  };
  // 512 threads at most: the state, the candidate and seven slopes of a vector are 36 registers, and a 1024-thread
  // workgroup's 128 per thread spilled even with one vector each
  if (kmax <= 1) return go(small_tsit5_kernel<T, EQ, CL, 1, 512>);
  if (kmax <= 2) return go(small_tsit5_kernel<T, EQ, CL, 2, 512>);
  return go(small_tsit5_kernel<T, EQ, CL, 4, 512>);
}

// The whole adaptive solve t0 -> t1 of every environment, one launch.  save_host: [n_save][batch][nx][ny].
template <typename T>
int small_tsit5_solve(pdeopt_ctx* ctx, double t0, double t1, double dt0, const pdeopt_pid* pid, int64_t max_steps, int n_save,
                      const double* save_ts, void* save_host, pdeopt_tsit5_stats* stats_host) {
  constexpr int V = VecOf<T>::V;
  const pdeopt_problem& p = ctx->prob;
  const int batch = p.batch;
  const int64_t cells = (int64_t)p.nx * p.ny;
  SmallTsit5Args<T> s{};
  const Geo g = make_geo(ctx);
  s.y = static_cast<T*>(ctx->Y);
  const SmallDims d = small_dims(p);
  s.nx = d.nx;
  s.ny = d.ny;
  s.bstride = g.bstride;
  s.rhx = T(0.5 * d.rx2); s.rhy = T(0.5 * d.ry2);
  s.rhx2 = T(d.rx2); s.rhy2 = T(d.ry2);
  s.ep = static_cast<const EnvParams<T>*>(ctx->env_params_dev);
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  s.t0 = t0; s.t1 = t1; s.dt0 = dt0;
  const double order = 5.0;
  s.pid.rtol = pid->rtol; s.pid.atol = pid->atol;
  s.pid.k1 = (pid->icoeff + pid->pcoeff + pid->dcoeff) / order;
  s.pid.k2 = -(pid->pcoeff + 2 * pid->dcoeff) / order;
  s.pid.k3 = pid->dcoeff / order;
  s.pid.factormin = pid->factormin; s.pid.factormax = pid->factormax; s.pid.safety = pid->safety;
  s.pid.dtmin = pid->dtmin; s.pid.dtmax = pid->dtmax;
  s.max_steps = max_steps;
  s.n_save = n_save;
  s.save_stride = (int64_t)batch * cells;

  // one device block for the save times, the statistics and the save slots
  const size_t ts_bytes = ((size_t)n_save * sizeof(double) + 255) / 256 * 256;
  const size_t st_bytes = ((size_t)batch * sizeof(pdeopt_tsit5_stats) + 255) / 256 * 256;
  const size_t out_bytes = (size_t)n_save * batch * cells * sizeof(T);
  // kept with the ctx (grow-only): a solve of a 64^2 grid is a few hundred microseconds, a hipMalloc / hipFree pair is not free
  const size_t need = ts_bytes + st_bytes + out_bytes + 256;
  if (ctx->adaptive_cap < need) {
    if (ctx->adaptive_blk) {
      PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->adaptive_blk);
      ctx->adaptive_blk = nullptr;
      ctx->adaptive_cap = 0;
    }
    PDEOPT_HIP_CHECK(ctx, hipMalloc(&ctx->adaptive_blk, need));
    ctx->adaptive_cap = need;
  }
  char* const blk = static_cast<char*>(ctx->adaptive_blk);
  s.save_ts = reinterpret_cast<const double*>(blk);
  s.stats = reinterpret_cast<pdeopt_tsit5_stats*>(blk + ts_bytes);
  s.save_out = reinterpret_cast<T*>(blk + ts_bytes + st_bytes);
  if (n_save) {
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(blk, save_ts, (size_t)n_save * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipMemsetAsync(s.save_out, 0xFF, out_bytes, ctx->stream));  // all-ones words: NaN in fp32 and fp64
  }
  PDEOPT_HIP_CHECK(ctx, hipMemsetAsync(s.stats, 0, st_bytes, ctx->stream));

  const int64_t nvec = (int64_t)d.nx * (d.ny / V);
  int nt, kmax;
  if (nvec <= 512) { nt = (int)((nvec + 63) / 64 * 64); kmax = 1; }
  else { nt = 512; kmax = (int)((nvec + 511) / 512); }
  // [input A][mu][input B]([state]): small_tsit5_kernel.  The state array belongs to the KMAX = 4 INSTANTIATION, which
  // also runs 3 vectors per thread (launch_small_tsit5_k): sized by kmax >= 4 a 64 x 96 grid ran with its state array
  // past the allocation (round 4: found by the 64 x 96 / 48 x 48 fp64 shapes ADVICE r3 asked for)
  const size_t tile_bytes = ((size_t)cells * sizeof(T) * (kmax >= 3 ? 4 : 3) + 15) / 16 * 16;
  s.red_off = (int)tile_bytes;
  const size_t lds = tile_bytes + 16 * sizeof(double);
  const int cl = classify_closures(p.mu, p.mob);
  char name[96];
  snprintf(name, sizeof(name), "small_tsit5<%s,%s,%s,%d threads,%d vec/thread>", sizeof(T) == 4 ? "f32" : "f64",
           p.equation == PDEOPT_EQ_ALLEN_CAHN ? "AC" : "CH", cl == CL_LOGIT ? "logit" : "poly", nt, kmax);
  ctx->last_kernel = name;
  ctx->n_stage_launches++;
  ctx->tsit5_pending = false;
  ctx->tsit5_fsal_valid = false;
  int rc;
  if (p.equation == PDEOPT_EQ_CAHN_HILLIARD) {
    if (cl == CL_LOGIT && p.mu.n <= 2) rc = launch_small_tsit5_k<T, PDEOPT_EQ_CAHN_HILLIARD, CL_LOGIT1>(ctx, s, nt, kmax, lds);
    else if (cl == CL_LOGIT) rc = launch_small_tsit5_k<T, PDEOPT_EQ_CAHN_HILLIARD, CL_LOGIT>(ctx, s, nt, kmax, lds);
    else rc = launch_small_tsit5_k<T, PDEOPT_EQ_CAHN_HILLIARD, CL_POLY>(ctx, s, nt, kmax, lds);
  } else if (cl == CL_LOGIT) {
    rc = launch_small_tsit5_k<T, PDEOPT_EQ_ALLEN_CAHN, CL_LOGIT>(ctx, s, nt, kmax, lds);
  } else {
    rc = launch_small_tsit5_k<T, PDEOPT_EQ_ALLEN_CAHN, CL_POLY>(ctx, s, nt, kmax, lds);
  }
  if (rc) return rc;
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(stats_host, s.stats, (size_t)batch * sizeof(pdeopt_tsit5_stats), hipMemcpyDeviceToHost, ctx->stream));
  if (n_save) PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(save_host, s.save_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PDEOPT_OK;
}

}  // namespace pdeopt
