// Host-side dispatch of the fused stencil + integrator-stage kernels (explicit integrators).
//
// Replaces the body of diffrax.diffeqsolve's while-loop for explicit solvers at the call sites
// pde_opt/pde_env.py:293-303 and pde_opt/pde_model.py:120-134: one kernel per Runge-Kutta stage,
// each evaluating equation.rhs on the stage input and applying the stage's axpy updates in the
// same pass.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "stencil_generic.hpp"
#include "stencil_tiled.hpp"
#include "stencil_fused.hpp"
#include "stencil_fused_ac.hpp"
#include "stencil_fused_ac4.hpp"
#include "stencil_fused_ch4.hpp"
#include "stencil_fused_launch.hpp"
#include "stencil_small.hpp"
#include "stencil_small_adaptive.hpp"
#include "stencil_coop_adaptive.hpp"
#include "stencil_sbm_tiled.hpp"

namespace pdeopt {

namespace {

template <typename T>
StageArgs<T> make_args(pdeopt_ctx* ctx, const void* in, const void* y, void* out, void* acc,
                       double a, double b, int out_mode, int acc_mode) {
  const pdeopt_problem& p = ctx->prob;
  StageArgs<T> s{};
  // environment window [win_lo, win_lo + win_n): pointers are pre-offset, kernels see a batch of win_n
  s.g = make_geo(ctx);
  const int64_t woff = (int64_t)ctx->win_lo * s.g.bstride;
  s.in = static_cast<const T*>(in) + woff;
  s.y = static_cast<const T*>(y) + woff;
  s.out = out ? static_cast<T*>(out) + woff : nullptr;
  s.acc = acc ? static_cast<T*>(acc) + woff : nullptr;
  s.a = T(a);
  s.b = T(b);
  s.rhx = T(1.0 / p.hx);
  s.rhy = T(1.0 / p.hy);
  s.rhx2 = T(1.0 / (p.hx * p.hx));
  s.rhy2 = T(1.0 / (p.hy * p.hy));
  s.rhz = p.nz > 1 ? T(1.0 / p.hz) : T(0);
  s.rhz2 = p.nz > 1 ? T(1.0 / (p.hz * p.hz)) : T(0);
  s.mu3 = nullptr;
  s.ep = static_cast<const EnvParams<T>*>(ctx->env_params_dev) + ctx->win_lo;
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  s.vstride = ctx->aux[PDEOPT_AUX_VX_FACE].per_env ? (int64_t)p.nx * p.ny : 0;
  s.vx = static_cast<const T*>(ctx->aux[PDEOPT_AUX_VX_FACE].dev);
  s.vy = static_cast<const T*>(ctx->aux[PDEOPT_AUX_VY_FACE].dev);
  if (s.vx) s.vx += ctx->win_lo * s.vstride;
  if (s.vy) s.vy += ctx->win_lo * s.vstride;
  s.psi = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_PSI].dev);
  s.ngp = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_NORM_GRAD].dev);
  s.mask = static_cast<const T*>(ctx->aux[PDEOPT_AUX_SBM_MASK].dev);
  s.fe = ClosureSpec{p.fe.kind, p.fe.flags, p.fe.n};
  if (p.equation == PDEOPT_EQ_ALLEN_CAHN_SBM || p.equation == PDEOPT_EQ_CAHN_HILLIARD_SBM) {
    double tv[3] = {ctx->time_const[0], ctx->time_const[1], ctx->time_const[2]};
    // the table of pdeopt_set_time_table first (stage times arrive in order: resume at the cursor), the host callback
    // for times it does not hold
    bool found = false;
    const size_t nt = ctx->tt_times.size();
    for (size_t q = 0; q < nt && !found; ++q) {
      const size_t i = (ctx->tt_cursor + q) % nt;
      if (ctx->tt_times[i] == ctx->cur_t) {
        for (int c = 0; c < 3; ++c) tv[c] = ctx->tt_terms[3 * i + c];
        ctx->tt_cursor = i;
        found = true;
      }
    }
    if (!found && ctx->time_fn) ctx->time_fn(ctx->cur_t, tv, ctx->time_user);
    // a table that does not hold this stage time and no callback to ask: the constant terms would be used silently
    // (a caller whose times differ from the library's t0 + s dt, + dt/2 arithmetic by an ulp) -- pdeopt_advance reports it
    if (!found && !ctx->time_fn && nt) ctx->tt_misses++;
    s.tw_a = T(tv[0]);
    s.tw_b = T(tv[1]);
    s.tsrc = T(tv[2]);
  }
  s.out_mode = out_mode;
  s.acc_mode = acc_mode;
  s.scaled = ctx->slope_scaled ? 1 : 0;
  s.dbg = (int)ctx->opt_debug_ablate;
  return s;
}

template <typename T>
int launch_generic(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  const pdeopt_problem& p = ctx->prob;
  dim3 block(64, 4, 1);
  dim3 grid((p.ny + 63) / 64, (p.nx + 3) / 4, ctx->win_n);
  if (grid.y > 65535u || grid.z > 65535u)
    return fail(ctx, PDEOPT_EINVAL, "grid too large for the generic kernel (nx=%d batch=%d)", p.nx,
                p.batch);
  if (jit_closures_active(ctx)) return launch_jit_stage<T>(ctx, s);  // closures compiled at run time (jit.hip)
  if (p.equation == PDEOPT_EQ_CAHN_HILLIARD_3D) {
    // two passes: chemical potential into the work field, then the flux divergence + stage update
    const int nz = s.g.nz;
    dim3 g3((nz + 63) / 64, (p.ny + 3) / 4, (unsigned)(p.nx * ctx->win_n));
    if (g3.y > 65535u || g3.z > 65535u) return fail(ctx, PDEOPT_EINVAL, "grid too large for the 3-D kernels");
    int rc = ensure_buffer(ctx, &ctx->KS, ctx->total_bytes);
    if (rc) return rc;
    StageArgs<T> s3 = s;
    s3.mu3 = static_cast<const T*>(ctx->KS) + (int64_t)ctx->win_lo * s.g.bstride;
    switch (classify_closures(p.mu, p.mob)) {
      case CL_POLY:
        hipLaunchKernelGGL((ch3d_mu_kernel<T, CL_POLY>), g3, block, 0, ctx->stream, s3, const_cast<T*>(s3.mu3));
        hipLaunchKernelGGL((ch3d_stage_kernel<T, CL_POLY>), g3, block, 0, ctx->stream, s3);
        ctx->last_kernel = "stage_generic<CH-3D,poly>";
        break;
      case CL_LOGIT:
        hipLaunchKernelGGL((ch3d_mu_kernel<T, CL_LOGIT>), g3, block, 0, ctx->stream, s3, const_cast<T*>(s3.mu3));
        hipLaunchKernelGGL((ch3d_stage_kernel<T, CL_LOGIT>), g3, block, 0, ctx->stream, s3);
        ctx->last_kernel = "stage_generic<CH-3D,logit>";
        break;
      default:
        hipLaunchKernelGGL((ch3d_mu_kernel<T, CL_GENERIC>), g3, block, 0, ctx->stream, s3, const_cast<T*>(s3.mu3));
        hipLaunchKernelGGL((ch3d_stage_kernel<T, CL_GENERIC>), g3, block, 0, ctx->stream, s3);
        ctx->last_kernel = "stage_generic<CH-3D>";
    }
    PDEOPT_HIP_CHECK(ctx, hipGetLastError());
    return PDEOPT_OK;
  }
  switch (p.equation) {
    case PDEOPT_EQ_CAHN_HILLIARD:
      hipLaunchKernelGGL((stage_generic_kernel<T, PDEOPT_EQ_CAHN_HILLIARD>), grid, block, 0,
                         ctx->stream, s);
      ctx->last_kernel = "stage_generic<CH>";
      break;
    case PDEOPT_EQ_ALLEN_CAHN:
      hipLaunchKernelGGL((stage_generic_kernel<T, PDEOPT_EQ_ALLEN_CAHN>), grid, block, 0,
                         ctx->stream, s);
      ctx->last_kernel = "stage_generic<AC>";
      break;
    case PDEOPT_EQ_ADVECTION_DIFFUSION:
      if (!s.vx || !s.vy)
        return fail(ctx, PDEOPT_ESTATE, "advection-diffusion needs VX_FACE and VY_FACE aux fields");
      hipLaunchKernelGGL((stage_generic_kernel<T, PDEOPT_EQ_ADVECTION_DIFFUSION>), grid, block, 0,
                         ctx->stream, s);
      ctx->last_kernel = "stage_generic<AD>";
      break;
    case PDEOPT_EQ_SHAPE_SMOOTH:
      if (ctx->halo) return fail(ctx, PDEOPT_EINVAL, "shape smoothing needs the periodic layout");
      hipLaunchKernelGGL((stage_generic_kernel<T, PDEOPT_EQ_SHAPE_SMOOTH>), grid, block, 0, ctx->stream, s);
      ctx->last_kernel = "stage_generic<shape-smooth>";
      break;
    case PDEOPT_EQ_ALLEN_CAHN_SBM:
    case PDEOPT_EQ_CAHN_HILLIARD_SBM:
      if (!s.psi || !s.ngp || !s.mask)
        return fail(ctx, PDEOPT_ESTATE, "smoothed-boundary equations need the SBM_PSI, SBM_NORM_GRAD and SBM_MASK aux fields");
      if (ctx->halo) return fail(ctx, PDEOPT_EINVAL, "smoothed-boundary equations need the periodic layout");
      if (sbm_tiled_supported<T>(ctx)) {
        const int rc_t = launch_sbm_tiled<T>(ctx, s);
        if (rc_t) return rc_t;
        break;
      }
      if (p.equation == PDEOPT_EQ_ALLEN_CAHN_SBM) {
        hipLaunchKernelGGL((stage_generic_kernel<T, PDEOPT_EQ_ALLEN_CAHN_SBM>), grid, block, 0, ctx->stream, s);
        ctx->last_kernel = "stage_generic<AC-SBM>";
      } else if (ctx->opt_fuse_stages < 0 ||
                 (ctx->opt_fuse_stages == 0 && (int64_t)p.nx * p.ny * ctx->win_n < (1 << 18))) {
        // the literal one-pass form (inner re-evaluated at 5 points per cell): one launch instead of two, faster
        // on the notebook-sized grids (128^2: 7.0 vs 7.7 us per evaluation); PDEOPT_OPT_FUSE_STAGES = 1 / -1
        // force the two-pass / the literal form
        hipLaunchKernelGGL((stage_generic_kernel<T, PDEOPT_EQ_CAHN_HILLIARD_SBM>), grid, block, 0, ctx->stream, s);
        ctx->last_kernel = "stage_generic<CH-SBM>";
      } else {
        // two passes: inner once per cell into the work field, then the psi-weighted flux divergence
        // (1024^2 fp32: 24.8 vs 30.6 us per evaluation; both forms are L2-bound one-thread-per-cell kernels)
        int rc = ensure_buffer(ctx, &ctx->KS, ctx->total_bytes);
        if (rc) return rc;
        StageArgs<T> s2 = s;
        s2.mu3 = static_cast<const T*>(ctx->KS) + (int64_t)ctx->win_lo * s.g.bstride;
        hipLaunchKernelGGL(sbm_inner_kernel<T>, grid, block, 0, ctx->stream, s2, const_cast<T*>(s2.mu3));
        hipLaunchKernelGGL(sbm_ch_stage_kernel<T>, grid, block, 0, ctx->stream, s2);
        ctx->last_kernel = "stage_two_pass<CH-SBM>";
      }
      break;
    default:
      return fail(ctx, PDEOPT_EINVAL, "equation %d has no explicit RHS kernel", p.equation);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

// pointwise stage update for RHS evaluations that are not fused into a stencil kernel
template <typename T>
__global__ void stage_update_kernel(const StageArgs<T> a, const T* __restrict__ k, int64_t total, int64_t env_elems) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += st) stage_update<T>(a, i, a.scaled ? k[i] * a.ep[i / env_elems].kscale : k[i]);
}

// field of environment b *= ratio[b]
template <typename T>
__global__ void env_scale_kernel(T* __restrict__ f, const double* __restrict__ ratio, int64_t env_elems) {
  const int b = blockIdx.y;
  const T r = (T)ratio[b];
  T* p = f + (int64_t)b * env_elems;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < env_elems; i += (int64_t)gridDim.x * blockDim.x) p[i] *= r;
}

template <typename T>
int launch_stage_fourier(pdeopt_ctx* ctx, const StageArgs<T>& s, const void* in, void* out) {
  if (ctx->win_lo != 0 || ctx->win_n != ctx->prob.batch)
    return fail(ctx, PDEOPT_EINVAL, "spectral RHS works on the whole batch");
  int rc;
  if (s.out_mode == OUT_K && s.acc_mode == ACC_NONE && !s.scaled) return rhs_fourier(ctx, in, out);
  if ((rc = ensure_buffer(ctx, &ctx->KS, ctx->total_bytes))) return rc;
  if ((rc = rhs_fourier(ctx, in, ctx->KS))) return rc;
  const int64_t total = (int64_t)(ctx->env_elems * ctx->prob.batch);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(stage_update_kernel<T>, dim3(blocks), dim3(256), 0, ctx->stream, s, (const T*)ctx->KS, total,
                     (int64_t)ctx->env_elems);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

// One fused stage: k = rhs(in); out/acc updated per (out_mode, acc_mode).
template <typename T>
int launch_stage_t(pdeopt_ctx* ctx, const void* in, const void* y, void* out, void* acc, double a,
                   double b, int out_mode, int acc_mode) {
  if (ctx->prob.equation == PDEOPT_EQ_ADVECTION_DIFFUSION) {
    // velocity_fn(t, x, y): face velocities at the time of THIS right-hand side (ctx->cur_t = stage time)
    int rc;
    if ((rc = refresh_time_aux(ctx, PDEOPT_AUX_VX_FACE, ctx->cur_t))) return rc;
    if ((rc = refresh_time_aux(ctx, PDEOPT_AUX_VY_FACE, ctx->cur_t))) return rc;
  }
  StageArgs<T> s = make_args<T>(ctx, in, y, out, acc, a, b, out_mode, acc_mode);
  ctx->n_stage_launches++;
  if (ctx->prob.derivs == PDEOPT_DERIVS_FOURIER) return launch_stage_fourier<T>(ctx, s, in, out);
  if (ctx->opt_kernel_path != 1 && tiled_supported<T>(ctx)) {
    return launch_tiled<T>(ctx, s);
  }
  if (ctx->opt_kernel_path == 2)
    return fail(ctx, PDEOPT_EINVAL, "LDS-tiled kernel forced but shape %dx%d / equation %d is not covered",
                ctx->prob.nx, ctx->prob.ny, ctx->prob.equation);
  return launch_generic<T>(ctx, s);
}

int launch_stage(pdeopt_ctx* ctx, const void* in, const void* y, void* out, void* acc, double a,
                 double b, int out_mode, int acc_mode) {
  if (ctx->prob.dtype == PDEOPT_F32)
    return launch_stage_t<float>(ctx, in, y, out, acc, a, b, out_mode, acc_mode);
  return launch_stage_t<double>(ctx, in, y, out, acc, a, b, out_mode, acc_mode);
}

// k = rhs(in) -> kout, and  next = y + sum_{j<n} c[j] K[j] + c[n] k  in the same pass (OUT_K_LC)
template <typename T>
int launch_stage_lc(pdeopt_ctx* ctx, const void* in, const void* y, void* kout, void* const* ks, const double* c,
                    int n, void* next) {
  if (ctx->prob.equation == PDEOPT_EQ_ADVECTION_DIFFUSION) {
    int rc;
    if ((rc = refresh_time_aux(ctx, PDEOPT_AUX_VX_FACE, ctx->cur_t))) return rc;
    if ((rc = refresh_time_aux(ctx, PDEOPT_AUX_VY_FACE, ctx->cur_t))) return rc;
  }
  StageArgs<T> s = make_args<T>(ctx, in, y, kout, nullptr, 0.0, 0.0, OUT_K_LC, ACC_NONE);
  const int64_t woff = (int64_t)ctx->win_lo * s.g.bstride;
  for (int j = 0; j < n; ++j) {
    s.lc.k[j] = static_cast<const T*>(ks[j]) + woff;
    s.lc.c[j] = T(c[j]);
  }
  s.lc.c[n] = T(c[n]);
  s.lc.n = n;
  s.lc.next = static_cast<T*>(next) + woff;
  ctx->n_stage_launches++;
  if (ctx->prob.derivs == PDEOPT_DERIVS_FOURIER) return launch_stage_fourier<T>(ctx, s, in, kout);
  if (ctx->opt_kernel_path != 1 && tiled_supported<T>(ctx)) return launch_tiled<T>(ctx, s);
  return launch_generic<T>(ctx, s);
}

int launch_pair_dt(pdeopt_ctx* ctx, int pair, const void* in, const void* y, const void* acc, void* out,
                   void* acc_out, double aA, double bA, double aB, double bB) {
  if (ctx->prob.dtype == PDEOPT_F32)
    return launch_pair<float>(ctx, pair, in, y, acc, out, acc_out, aA, bA, aB, bB);
  return launch_pair<double>(ctx, pair, in, y, acc, out, acc_out, aA, bA, aB, bB);
}

template <typename T>
__global__ void lerp_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                            T theta, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = a[i] + theta * (b[i] - a[i]);
}

}  // namespace

int launch_rhs(pdeopt_ctx* ctx, const void* in, void* out, double t) {
  ctx->cur_t = t;
  return launch_stage(ctx, in, in, out, nullptr, 0.0, 0.0, OUT_K, ACC_NONE);
}

#ifndef PDEOPT_IMEX_SLOPE_PAIR
#define PDEOPT_IMEX_SLOPE_PAIR 1
#endif
int launch_rhs_slope(pdeopt_ctx* ctx, const void* in, void* out, double t) {
  ctx->cur_t = t;
#if PDEOPT_IMEX_SLOPE_PAIR
  if (ctx->prob.derivs == PDEOPT_DERIVS_FD && ctx->prob.nz <= 1 && !ctx->time_fn) {
    if (ctx->prob.dtype == PDEOPT_F32 && slope_pair_supported<float>(ctx)) return launch_slope_pair<float>(ctx, in, out);
    if (ctx->prob.dtype == PDEOPT_F64 && slope_pair_supported<double>(ctx)) return launch_slope_pair<double>(ctx, in, out);
  }
#endif
  return launch_rhs(ctx, in, out, t);
}

namespace {
constexpr int kGraphUnit = 16;  // substeps per captured graph (even: ping-pong buffers return)

// the part of a problem that is baked into kernel arguments (not the per-env parameter values)
GraphStructure graph_structure(const pdeopt_problem& p) {
  GraphStructure g{};
  g.equation = p.equation; g.dtype = p.dtype; g.nx = p.nx; g.ny = p.ny; g.batch = p.batch; g.derivs = p.derivs;
  g.nz = p.nz; g.hx = p.hx; g.hy = p.hy; g.hz = p.hz;
  g.mu_kind = p.mu.kind; g.mu_flags = p.mu.flags; g.mu_n = p.mu.n;
  g.mob_kind = p.mob.kind; g.mob_flags = p.mob.flags; g.mob_n = p.mob.n;
  return g;
}
}  // namespace

// Does the whole-environment-step kernel (stencil_small.hpp) take this advance?  PDEOPT_OPT_SMALL_PERSIST: 1 = wherever
// it can run, -1 = never, 0 = auto: grids of at most 64^2-class size (<= 4096 cells: one launch instead of 2 n
// dependent ones, measured 2.7-4.8 x faster per environment step), and larger LDS-resident grids once enough
// environments run side by side to fill the chip's compute units (>= 192) -- a single 128^2 environment is faster on
// the tiled kernels, which spread it over 8+ CUs, than on one CU.
static bool small_chosen(const pdeopt_ctx* ctx, int integrator, int64_t n) {
  if (ctx->opt_small_persist < 0 || ctx->opt_kernel_path == 1 || ctx->opt_debug_ablate) return false;
  if (integrator != PDEOPT_INT_EULER && integrator != PDEOPT_INT_RK4) return false;
  const bool ok = ctx->prob.dtype == PDEOPT_F32 ? small_supported<float>(ctx) : small_supported<double>(ctx);
  if (!ok) return false;
  if (ctx->opt_small_persist > 0) return true;
  // a caller who turned one of the tiled path's knobs is asking for that path
  if (ctx->opt_fuse_stages != 0 || ctx->opt_kernel_path != 0 || ctx->opt_graph != 0 || ctx->opt_group_envs != 0 || ctx->opt_tile_rows != 0)
    return false;
  const int64_t cells = (int64_t)ctx->prob.nx * ctx->prob.ny;
  if (n < 2) return false;
  return cells <= kSmallAutoCells || ctx->prob.batch >= kSmallAutoBatch;
}

int advance_explicit(pdeopt_ctx* ctx, int integrator, double t0, double dt, int64_t n) {
  int rc;
  // one environment (or a few) of a mid-sized grid: several compute units per environment, all substeps in one launch
  // (stencil_coop_adaptive.hpp, MODE 1)
  const bool coop = ctx->prob.dtype == PDEOPT_F32 ? coop_fixed_chosen<float>(ctx, integrator, n) : coop_fixed_chosen<double>(ctx, integrator, n);
  const bool f64 = ctx->prob.dtype == PDEOPT_F64;  // (64^2 fp64: 0.70 ms against the one-CU kernel's 1.14, which small_chosen would pick)
  if (coop && (ctx->opt_small_persist == 2 || f64 || !small_chosen(ctx, integrator, n))) {
    ctx->win_lo = 0;
    ctx->win_n = ctx->prob.batch;
    ctx->last_groups = 1;
    return ctx->prob.dtype == PDEOPT_F32 ? coop_fixed_advance<float>(ctx, integrator, t0, dt, n) : coop_fixed_advance<double>(ctx, integrator, t0, dt, n);
  }
  if (small_chosen(ctx, integrator, n)) {
    ctx->win_lo = 0;
    ctx->win_n = ctx->prob.batch;
    ctx->last_groups = 1;
    return ctx->prob.dtype == PDEOPT_F32 ? launch_small<float>(ctx, integrator, dt, n) : launch_small<double>(ctx, integrator, dt, n);
  }
  if ((rc = ensure_buffer(ctx, &ctx->TA, ctx->total_bytes))) return rc;
  // Euler: two substeps per launch through the stage-pair kernels where they exist
  //   PAIR_12 with aA = bA = bB = dt:  w = y + dt f(y),  ACC = y + dt f(y) + dt f(w) = two Euler steps
  const bool euler2 = integrator == PDEOPT_INT_EULER && ctx->opt_kernel_path != 1 &&
                      ctx->prob.derivs == PDEOPT_DERIVS_FD && n >= 2 &&
                      (ctx->prob.dtype == PDEOPT_F32 ? fused_supported<float>(ctx) : fused_supported<double>(ctx));
  if (integrator == PDEOPT_INT_RK4 || euler2) {
    if ((rc = ensure_buffer(ctx, &ctx->TB, ctx->total_bytes))) return rc;
    if (integrator == PDEOPT_INT_RK4 && (rc = ensure_buffer(ctx, &ctx->ACC, ctx->total_bytes))) return rc;
  } else if (integrator != PDEOPT_INT_EULER) {
    return fail(ctx, PDEOPT_EINVAL, "integrator %d is not an explicit fixed-step integrator", integrator);
  }
  // work field of the two-pass 3-D right-hand side: allocated here, never inside a stream capture
  if (ctx->prob.equation == PDEOPT_EQ_CAHN_HILLIARD_3D && (rc = ensure_buffer(ctx, &ctx->KS, ctx->total_bytes)))
    return rc;
  // Environments are independent, so the substep loop may run group by group: a group whose
  // working set (4 fields x group x nx x ny) fits the 256 MiB Infinity Cache keeps every stage's
  // reads and writes on-die for all n substeps instead of streaming the whole batch through HBM
  // once per stage.
  const int batch = ctx->prob.batch;
  // time-dependent terms evaluated by the host per stage (smoothed-boundary scalars, advection velocities)
  const bool timed = ctx->prob.equation == PDEOPT_EQ_ALLEN_CAHN_SBM || ctx->prob.equation == PDEOPT_EQ_CAHN_HILLIARD_SBM ||
                     has_time_aux(ctx, PDEOPT_AUX_VX_FACE) || has_time_aux(ctx, PDEOPT_AUX_VY_FACE);
  int group = batch;
  bool side_by_side = false;
  if (ctx->prob.derivs == PDEOPT_DERIVS_FOURIER || timed) {
    group = batch;  // batched rocFFT plans cover the whole batch; host callbacks run once per stage time
  } else if (ctx->opt_group_envs > 0) {
    group = (int)std::min<int64_t>(ctx->opt_group_envs, batch);
  } else if (ctx->opt_group_envs == 0) {
    // auto: largest balanced group whose 4 fields fit the Infinity Cache (measured on MI355X:
    // 32 x 1024^2 fp32 runs 13-16 % faster as 2 groups of 16 = 256 MiB than as one 512 MiB sweep;
    // smaller groups lose more to per-launch ramp/tail than they gain)
    const size_t field = ctx->env_elems * ctx->esize;
    const int64_t fit = std::max<int64_t>(1, (int64_t)((256ull << 20) / (4 * field)));
    if (fit < batch && n > 1) {
      const int ngroups = (int)((batch + fit - 1) / fit);
      group = (batch + ngroups - 1) / ngroups;
      // Two groups side by side on two streams, each half the size (the same resident working set): a launch of the
      // stage kernels is 3-6 rounds of resident workgroups, its ramp and its tail leave compute units idle, and the
      // next launch of the SAME group depends on it -- the other group's does not.
      if (ctx->opt_group_streams != 1 && group >= 2 && ctx->prob.equation != PDEOPT_EQ_CAHN_HILLIARD_3D) {
        group = (group + 1) / 2;
        side_by_side = true;
      }
    }
  }
  const bool can_pair = !timed && ctx->prob.derivs == PDEOPT_DERIVS_FD && ctx->prob.equation != PDEOPT_EQ_CAHN_HILLIARD_3D;
  if (ctx->opt_group_streams == 2 && can_pair && group < batch) side_by_side = true;
  // ... and a batch that fits the cache as one group runs as two halves side by side for the same reason (measured
  // 512^2 x 64 Allen-Cahn: 20.1 k env-steps/s against 18.9 k), unless it is small enough for the hipGraph replay below
  if (ctx->opt_group_streams == 0 && ctx->opt_group_envs == 0 && can_pair && group >= batch && batch >= 2 && n > 1 &&
      ctx->opt_graph <= 0 && (int64_t)ctx->prob.nx * ctx->prob.ny * batch > (1 << 21)) {
    group = (batch + 1) / 2;
    side_by_side = true;
  }
  const bool fused = integrator == PDEOPT_INT_RK4 && ctx->opt_kernel_path != 1 &&
                     ctx->prob.derivs == PDEOPT_DERIVS_FD &&
                     (ctx->prob.dtype == PDEOPT_F32 ? fused_supported<float>(ctx) : fused_supported<double>(ctx));

  // Allen-Cahn fp32: the whole RK4 substep in one pass (2 words per cell instead of 7)
  const bool quad = integrator == PDEOPT_INT_RK4 && ac_quad_supported(ctx);
  // Cahn-Hilliard fp32, periodic divisible grids: the same (stencil_fused_ch4.hpp); PDEOPT_OPT_FUSE_STAGES = 1 keeps the stage pairs
  const bool chquad = integrator == PDEOPT_INT_RK4 && ch_quad_chosen(ctx);

  // two Euler substeps in one launch (result into TA, TB takes the kernel's unused y + dt k2 output)
  auto euler_pair = [&](void*& Y, void*& TA) -> int {
    const int r = launch_pair_dt(ctx, PAIR_12, Y, nullptr, nullptr, ctx->TB, TA, dt, dt, dt, dt);
    std::swap(Y, TA);
    return r;
  };

  // one substep on the current window; Y / TA are swapped where the integrator ping-pongs
  auto substep = [&](void*& Y, void*& TA, int64_t step) -> int {
    int r;
    const double ts = t0 + (double)step * dt;  // stage times only matter to time-dependent equations
    ctx->cur_t = ts;
    if (integrator == PDEOPT_INT_EULER) {
      r = launch_stage(ctx, Y, Y, TA, nullptr, dt, 0.0, OUT_Y_PLUS_AK, ACC_NONE);
      std::swap(Y, TA);
      return r;
    }
    if (quad) {
      r = launch_ac_quad(ctx, Y, TA, dt);
      std::swap(Y, TA);
      return r;
    }
    if (chquad) {
      r = launch_ch_quad(ctx, Y, TA, dt);
      std::swap(Y, TA);
      return r;
    }
    if (fused) {
      // stages 1+2 and 3+4 as two temporally fused launches (7 words/cell instead of 16)
      r = launch_pair_dt(ctx, PAIR_12, Y, nullptr, nullptr, ctx->TB, ctx->ACC, dt / 2, dt / 6, dt / 2, dt / 3);
      if (!r) r = launch_pair_dt(ctx, PAIR_34, ctx->TB, Y, ctx->ACC, TA, nullptr, dt, dt / 3, 0.0, dt / 6);
      std::swap(Y, TA);
      return r;
    }
    // stage 1: k1 = f(y);        TA = y + dt/2 k1;  ACC = y + dt/6 k1
    r = launch_stage(ctx, Y, Y, TA, ctx->ACC, dt / 2, dt / 6, OUT_Y_PLUS_AK, ACC_INIT);
    // stage 2: k2 = f(TA);       TB = y + dt/2 k2;  ACC += dt/3 k2
    ctx->cur_t = ts + dt / 2;
    if (!r) r = launch_stage(ctx, TA, Y, ctx->TB, ctx->ACC, dt / 2, dt / 3, OUT_Y_PLUS_AK, ACC_ADD);
    // stage 3: k3 = f(TB);       TA = y + dt k3;    ACC += dt/3 k3
    if (!r) r = launch_stage(ctx, ctx->TB, Y, TA, ctx->ACC, dt, dt / 3, OUT_Y_PLUS_AK, ACC_ADD);
    // stage 4: k4 = f(TA);       y  = ACC + dt/6 k4   (in place: y is only touched pointwise)
    ctx->cur_t = ts + dt;
    if (!r) r = launch_stage(ctx, TA, Y, Y, ctx->ACC, 0.0, dt / 6, OUT_ACC_PLUS_BK, ACC_NONE);
    return r;
  };

  // Launch-bound regime (small grids / few environments: a stage kernel runs for a few microseconds,
  // the host needs ~3.5 us to issue one): capture kGraphUnit substeps once into a hipGraph and replay
  // it.  Kernel arguments baked into the graph are the field pointers, dt and the problem structure;
  // per-environment parameter VALUES live in device memory and may change between replays.
  int64_t done = 0;
  const int64_t cells_per_launch = (int64_t)ctx->prob.nx * ctx->prob.ny * group;
  const bool want_graph = !timed && ctx->opt_graph >= 0 && group >= batch && ctx->prob.derivs == PDEOPT_DERIVS_FD &&
                          n >= 2 * kGraphUnit && (ctx->opt_graph > 0 || cells_per_launch <= (1 << 20));
  if (want_graph) {
    ctx->win_lo = 0;
    ctx->win_n = batch;
    GraphKey key;
    memset(&key, 0, sizeof(key));  // padding bytes take part in the memcmp below
    key.integrator = integrator;
    key.fused = quad ? 100 : chquad ? 101 : (euler2 ? 50 : (fused ? (int)(1 + ctx->opt_fuse_stages) : 0));
    key.dt = dt;
    key.Y = ctx->Y; key.TA = ctx->TA; key.TB = ctx->TB; key.ACC = ctx->ACC; key.KS = ctx->KS;
    key.ep = ctx->env_params_dev;
    key.vx = ctx->aux[PDEOPT_AUX_VX_FACE].dev; key.vy = ctx->aux[PDEOPT_AUX_VY_FACE].dev;
    key.kernel_path = ctx->opt_kernel_path; key.tile_rows = ctx->opt_tile_rows; key.ablate = ctx->opt_debug_ablate;
    key.structure = graph_structure(ctx->prob);
    if (!ctx->graph_exec || memcmp(&key, &ctx->graph_key, sizeof(key)) != 0) {
      graph_destroy(ctx);
      void* Y = ctx->Y;
      void* TA = ctx->TA;
      const int64_t launches_before = ctx->n_stage_launches;
      hipGraph_t graph = nullptr;
      PDEOPT_HIP_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      for (int u = 0; u < kGraphUnit && !rc; ++u) {
        if (euler2) {
          rc = euler_pair(Y, TA);
          ++u;
        } else {
          rc = substep(Y, TA, u);
        }
      }
      const hipError_t e_end = hipStreamEndCapture(ctx->stream, &graph);
      ctx->graph_launches_per_replay = ctx->n_stage_launches - launches_before;
      ctx->n_stage_launches = launches_before;
      if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      PDEOPT_HIP_CHECK(ctx, e_end);
      const hipError_t e_inst = hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      PDEOPT_HIP_CHECK(ctx, e_inst);
      ctx->graph_key = key;
      ctx->graph_name = ctx->last_kernel + "+hipGraph";
      // kGraphUnit is even: the ping-pong buffers are back in place after one replay
    }
    for (; done + kGraphUnit <= n; done += kGraphUnit) {
      PDEOPT_HIP_CHECK(ctx, hipGraphLaunch(ctx->graph_exec, ctx->stream));
      ctx->n_stage_launches += ctx->graph_launches_per_replay;
    }
    ctx->last_kernel = ctx->graph_name;
  }

  void* y_final = ctx->Y;
  void* ta_final = ctx->TA;
  ctx->last_groups = (batch + group - 1) / group;
  if (side_by_side && ctx->last_groups >= 2) {
    ctx->last_group_streams = 2;
    if ((rc = ensure_stream2(ctx))) return rc;
    // the second stream starts after everything already queued on the ctx stream (the state upload, the previous call)
    PDEOPT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    for (int lo = 0; lo < batch && !rc; lo += 2 * group) {
      const int lo2 = lo + group;
      const bool two = lo2 < batch;
      void* Y = ctx->Y;
      void* TA = ctx->TA;
      for (int64_t s = done; s < n && !rc; ++s) {
        const bool pair = euler2 && s + 1 < n;
        void *Ya = Y, *TAa = TA, *Yb = Y, *TAb = TA;
        ctx->win_lo = lo;
        ctx->win_n = std::min(group, batch - lo);
        rc = pair ? euler_pair(Ya, TAa) : substep(Ya, TAa, s);
        if (two && !rc) {
          ctx->win_lo = lo2;
          ctx->win_n = std::min(group, batch - lo2);
          std::swap(ctx->stream, ctx->stream2);  // the launch helpers take the ctx stream
          rc = pair ? euler_pair(Yb, TAb) : substep(Yb, TAb, s);
          std::swap(ctx->stream, ctx->stream2);
        }
        Y = Ya;  // both groups rotate their buffers alike
        TA = TAa;
        if (pair) ++s;
      }
      y_final = Y;
      ta_final = TA;
    }
    const hipError_t e1 = hipEventRecord(ctx->ev_join, ctx->stream2);
    const hipError_t e2 = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);  // later work on the ctx stream sees both groups' results
    ctx->win_lo = 0;
    ctx->win_n = batch;
    if (rc) return rc;
    PDEOPT_HIP_CHECK(ctx, e1);
    PDEOPT_HIP_CHECK(ctx, e2);
    ctx->Y = y_final;
    ctx->TA = ta_final;
    return rc;
  }
  for (int lo = 0; lo < batch; lo += group) {
    ctx->win_lo = lo;
    ctx->win_n = std::min(group, batch - lo);
    void* Y = ctx->Y;
    void* TA = ctx->TA;
    for (int64_t s = done; s < n && !rc; ++s) {
      if (euler2 && s + 1 < n) {
        rc = euler_pair(Y, TA);
        ++s;
      } else {
        rc = substep(Y, TA, s);
      }
    }
    y_final = Y;  // every group performs the same number of swaps
    ta_final = TA;
    if (rc) break;
  }
  ctx->win_lo = 0;
  ctx->win_n = batch;
  ctx->Y = y_final;
  ctx->TA = ta_final;
  return rc;
}

void graph_destroy(pdeopt_ctx* ctx) {
  if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
  ctx->graph_exec = nullptr;
}

// One phase of an RK4 substep for callers that interleave their own work between phases (the
// domain-decomposed driver exchanges halos of `fields[phase]` before `rk4_phase(phase)`).
int rk4_phase_plan(pdeopt_ctx* ctx, int* fields, int* nphases) {
  const bool fused = ctx->opt_kernel_path != 1 && ctx->prob.derivs == PDEOPT_DERIVS_FD &&
                     (ctx->prob.dtype == PDEOPT_F32 ? fused_supported<float>(ctx) : fused_supported<double>(ctx));
  if (ctx->halo == 8 && ch_quad_chosen(ctx)) {
    // halo-8 layout, the whole substep in one kernel (stencil_fused_ch4.hpp): its tile + 8 input is the halo
    *nphases = 1;
    fields[0] = 0;
  } else if (fused && ctx->halo == 8 && ctx->prob.equation == PDEOPT_EQ_CAHN_HILLIARD) {
    // halo-8 layout: ONE exchange per substep.  Pair 1+2 runs on the tile + 4 ring (it reads y on tile + 8), so
    // pair 3+4 finds TB on its tile + 4 input region without an exchange of TB.
    *nphases = 2;
    fields[0] = 0;
    fields[1] = -1;  // no exchange before phase 1
  } else if (fused && ctx->halo != 8) {
    *nphases = 2;
    fields[0] = 0;  // Y  (stage pair 1+2 differentiates y)
    fields[1] = 2;  // TB (stage pair 3+4 differentiates y + dt/2 k2)
  } else {
    *nphases = 4;
    fields[0] = 0; fields[1] = 1; fields[2] = 2; fields[3] = 1;  // Y, TA, TB, TA
  }
  return PDEOPT_OK;
}

// part: 0 = the whole tile; 1 = interior tiles only, 2 = edge tiles only (fused stage pairs; the caller runs
// part 1 while the halo exchange of this phase is in flight and part 2 after pdeopt_halo_unpack).  The
// substep's buffer rotation happens with the LAST launch of the last phase (part 0 or 2).
int rk4_phase(pdeopt_ctx* ctx, int phase, double dt, int part) {
  if (ctx->prob.equation == PDEOPT_EQ_GPE) return fail(ctx, PDEOPT_EINVAL, "no explicit RHS for the GPE");
  if (part < 0 || part > 2) return fail(ctx, PDEOPT_EINVAL, "part %d outside 0..2", part);
  int rc;
  if ((rc = ensure_buffer(ctx, &ctx->TA, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &ctx->TB, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &ctx->ACC, ctx->total_bytes))) return rc;
  int fields[4], n = 0;
  rk4_phase_plan(ctx, fields, &n);
  if (phase < 0 || phase >= n) return fail(ctx, PDEOPT_EINVAL, "phase %d outside 0..%d", phase, n - 1);
  ctx->win_lo = 0;
  ctx->win_n = ctx->prob.batch;
  if (ctx->halo == 8 && n != 2 && n != 1)
    return fail(ctx, PDEOPT_EINVAL, "the halo-8 layout needs the fused Cahn-Hilliard kernels (closure class / tile shape / "
                                    "PDEOPT_OPT_FUSE_STAGES rule them out here): use halo layout 4");
  if (ctx->halo == 8 && part != 0)
    return fail(ctx, PDEOPT_EINVAL, "interior / edge launches belong to the halo-4 layout (two exchanges per substep)");
  if (n == 1) {  // the whole substep in one pass: y (+ halo) -> y'
    rc = launch_ch_quad(ctx, ctx->Y, ctx->TA, dt);
    std::swap(ctx->Y, ctx->TA);
    return rc;
  }
  if (n == 2) {
    ctx->launch_part = part;
    if (phase == 0) {
      ctx->pair_ext = ctx->halo == 8 ? 4 : 0;
      rc = launch_pair_dt(ctx, PAIR_12, ctx->Y, nullptr, nullptr, ctx->TB, ctx->ACC, dt / 2, dt / 6, dt / 2, dt / 3);
      ctx->pair_ext = 0;
    } else {
      rc = launch_pair_dt(ctx, PAIR_34, ctx->TB, ctx->Y, ctx->ACC, ctx->TA, nullptr, dt, dt / 3, 0.0, dt / 6);
      if (part != 1) std::swap(ctx->Y, ctx->TA);
    }
    ctx->launch_part = 0;
    return rc;
  }
  if (part != 0)
    return fail(ctx, PDEOPT_EINVAL, "interior / edge launches exist for the fused stage pairs only (this problem runs one kernel per stage)");
  switch (phase) {
    case 0: return launch_stage(ctx, ctx->Y, ctx->Y, ctx->TA, ctx->ACC, dt / 2, dt / 6, OUT_Y_PLUS_AK, ACC_INIT);
    case 1: return launch_stage(ctx, ctx->TA, ctx->Y, ctx->TB, ctx->ACC, dt / 2, dt / 3, OUT_Y_PLUS_AK, ACC_ADD);
    case 2: return launch_stage(ctx, ctx->TB, ctx->Y, ctx->TA, ctx->ACC, dt, dt / 3, OUT_Y_PLUS_AK, ACC_ADD);
    default: return launch_stage(ctx, ctx->TA, ctx->Y, ctx->Y, ctx->ACC, 0.0, dt / 6, OUT_ACC_PLUS_BK, ACC_NONE);
  }
}

// n RK4 substeps of a single-rank padded tile with the loop-back halo exchange, entirely in the library: the
// per-substep pack -> unpack -> phase sequence of the decomposed driver without a host round trip per call
// (what one rank of pde_opt_amd/decomp.py executes, minus the collective).
int rk4_loopback_advance(pdeopt_ctx* ctx, double dt, int64_t n) {
  int fields[4], np_ = 0;
  rk4_phase_plan(ctx, fields, &np_);
  const int nbr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int rc = PDEOPT_OK;
  if (ctx->halo == 8) {
    // one exchange per substep; after the first one the strip comes out of PAIR_34's epilogue (fused pack) and the
    // loop-back "collective" is the strip buffer itself
    if (n <= 0) return PDEOPT_OK;
    if ((rc = halo_pack(ctx, 0, nullptr))) return rc;
    // the strip written by substep s (fused pack) is read by substep s + 1 (fused unpack): two buffers
    const size_t bytes = halo_strip_elems(ctx) * ctx->esize;
    if (ctx->halo_scratch2_bytes < bytes) {  // its own grow-only size: ensure_buffer() keeps any existing pointer
      if (ctx->halo_scratch2) {
        PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->halo_scratch2);
        ctx->halo_scratch2 = nullptr;
        ctx->halo_scratch2_bytes = 0;
      }
      if ((rc = ensure_buffer(ctx, &ctx->halo_scratch2, bytes))) return rc;
      ctx->halo_scratch2_bytes = bytes;
    }
    void* cur = ctx->halo_scratch;
    void* nxt = ctx->halo_scratch2;
    for (int64_t s = 0; s < n && !rc; ++s) {
      rc = rk4_substep_h8(ctx, dt, s + 1 < n ? nxt : nullptr, cur, nbr);
      std::swap(cur, nxt);
    }
    return rc;
  }
  for (int64_t s = 0; s < n && !rc; ++s)
    for (int ph = 0; ph < np_ && !rc; ++ph) {
      if ((rc = halo_pack(ctx, fields[ph], nullptr))) break;
      if ((rc = halo_unpack(ctx, fields[ph], nullptr, nbr))) break;
      rc = rk4_phase(ctx, ph, dt, 0);
    }
  return rc;
}

// halo-8 layout, the state's halo already unpacked: both stage pairs of one substep; the edge tiles of the second
// write the NEW state's halo strip into `strip` (nullptr: not wanted)
// the same with the halo taken from the 8 neighbours' OWN strip buffers (peer-mapped exchange: comm.hip)
int rk4_substep_h8_peer(pdeopt_ctx* ctx, double dt, void* strip, const void* const* peer) {
  for (int q = 0; q < 8; ++q) ctx->pair_peer[q] = peer[q];
  const int rc = rk4_substep_h8(ctx, dt, strip, nullptr, nullptr);
  for (int q = 0; q < 8; ++q) ctx->pair_peer[q] = nullptr;
  return rc;
}

int rk4_substep_h8(pdeopt_ctx* ctx, double dt, void* strip, const void* recv, const int* nbr) {
  ctx->pair_recv = recv;
  if (recv)
    for (int q = 0; q < 8; ++q) ctx->pair_nbr[q] = nbr[q];
  if (ch_quad_chosen(ctx)) {  // one kernel: halo from the gathered strips in, the new strip out
    ctx->pair_strip = strip;
    const int rc = rk4_phase(ctx, 0, dt, 0);
    ctx->pair_recv = nullptr;
    ctx->pair_strip = nullptr;
    return rc;
  }
  int rc = rk4_phase(ctx, 0, dt, 0);
  ctx->pair_recv = nullptr;
  if (rc) return rc;
  ctx->pair_strip = strip;
  rc = rk4_phase(ctx, 1, dt, 0);
  ctx->pair_strip = nullptr;
  return rc;
}

// ------------------------------------------------------------------------------------------
// Tsit5 (diffrax.Tsit5; call sites tests/test_solvers.py:81,263 and most notebooks).  Tableau:
// Ch. Tsitouras, Comput. Math. Appl. 62 (2011) 770-775 -- the published coefficients diffrax
// ships.  Row f1 of SURVEY section 8: one trial step + scaled error norm per environment; the
// PID step-size logic stays on the host (pde_opt_amd/pde_model.py).
// ------------------------------------------------------------------------------------------
namespace {

template <typename T>
struct LinComb {
  const T* y;
  const T* k[7];
  T c[7];
  int n;
  T* out;
};

template <typename T>
__global__ void lincomb_kernel(const LinComb<T> a, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += st) {
    T r = a.y[i];
    for (int j = 0; j < a.n; ++j) r += a.c[j] * a.k[j][i];
    a.out[i] = r;
  }
}

constexpr int kErrBlocks = 64;

// partial[b][blk] = sum ((dt sum_j e_j k_j) / (atol + rtol max(|y0|,|y1|)))^2
template <typename T>
__global__ __launch_bounds__(256) void tsit5_err_kernel(const LinComb<T> a, const T* __restrict__ y1,
                                                        T rtol, T atol, int64_t env_elems,
                                                        double* __restrict__ partial) {
  const int b = blockIdx.y;
  const int64_t o = (int64_t)b * env_elems;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < env_elems; i += (int64_t)gridDim.x * 256) {
    T e = T(0);
    for (int j = 0; j < 7; ++j) e += a.c[j] * a.k[j][o + i];
    const T sc = atol + rtol * fmax(fabs(a.y[o + i]), fabs(y1[o + i]));
    const double q = (double)(e / sc);
    acc += q * q;
  }
  __shared__ double sh[4];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_down(acc, s, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

template <typename T>
int tsit5_trial_t(pdeopt_ctx* ctx, double t, double dt, double rtol, double atol, double* err) {
  int rc;
  for (auto& k : ctx->K)
    if ((rc = ensure_buffer(ctx, &k, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &ctx->TA, ctx->total_bytes))) return rc;
  if ((rc = ensure_buffer(ctx, &ctx->TB, ctx->total_bytes))) return rc;
  const int64_t total = (int64_t)(ctx->env_elems * ctx->prob.batch);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  if (!ctx->tsit5_fsal_valid) {
    ctx->cur_t = t;
    if ((rc = launch_stage(ctx, ctx->Y, ctx->Y, ctx->K[0], nullptr, 0, 0, OUT_K, ACC_NONE))) return rc;
  }
  // stage 2 input from k1 alone (k1 is the previous step's k7 under FSAL, so nothing could form it earlier)
  {
    LinComb<T> lc{};
    lc.y = (const T*)ctx->Y;
    lc.n = 1;
    lc.k[0] = (const T*)ctx->K[0];
    lc.c[0] = T(dt * kTsA[0][0]);
    lc.out = (T*)ctx->TA;
    hipLaunchKernelGGL(lincomb_kernel<T>, dim3(blocks), dim3(256), 0, ctx->stream, lc, total);
  }
  // stages 2..6: k_s = f(in_s) and, in the same pass, in_{s+1} = y + dt sum_j a_{s+1,j} k_j  (ping-pong
  // TA / TB; in_7 -- the 5th-order solution -- lands in TB).  Stage 7 evaluates k7 = f(in_7) for the
  // error estimate and the next step's FSAL.  8 launches per step instead of 13.
  for (int s = 1; s <= 5; ++s) {
    void* in_s = (s & 1) ? ctx->TA : ctx->TB;
    void* in_next = (s & 1) ? ctx->TB : ctx->TA;
    double c[kMaxLc + 1];
    for (int j = 0; j <= s; ++j) c[j] = dt * kTsA[s][j];
    ctx->cur_t = t + kTsC[s - 1] * dt;
    if ((rc = launch_stage_lc<T>(ctx, in_s, ctx->Y, ctx->K[s], ctx->K, c, s, in_next))) return rc;
  }
  ctx->cur_t = t + kTsC[5] * dt;
  if ((rc = launch_stage(ctx, ctx->TB, ctx->TB, ctx->K[6], nullptr, 0, 0, OUT_K, ACC_NONE))) return rc;
  ctx->tsit5_pending = true;
  if (err) {
    double* part = nullptr;
    const size_t need = sizeof(double) * kErrBlocks * ctx->prob.batch;
    if (ctx->red_cap < need) {
      if (ctx->red_dev) (void)hipFree(ctx->red_dev);
      ctx->red_dev = nullptr;
      PDEOPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->red_dev, need));
      ctx->red_cap = need;
    }
    part = ctx->red_dev;
    LinComb<T> lc{};
    lc.y = (const T*)ctx->Y;
    lc.n = 7;
    for (int j = 0; j < 7; ++j) {
      lc.k[j] = (const T*)ctx->K[j];
      lc.c[j] = T(dt * kTsE[j]);
    }
    hipLaunchKernelGGL(tsit5_err_kernel<T>, dim3(kErrBlocks, ctx->prob.batch), dim3(256), 0,
                       ctx->stream, lc, (const T*)ctx->TB, (T)rtol, (T)atol,
                       (int64_t)ctx->env_elems, part);
    PDEOPT_HIP_CHECK(ctx, hipGetLastError());
    std::vector<double> h((size_t)kErrBlocks * ctx->prob.batch);
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), part, need, hipMemcpyDeviceToHost, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < ctx->prob.batch; ++b) {
      double s = 0;
      for (int c = 0; c < kErrBlocks; ++c) s += h[(size_t)b * kErrBlocks + c];
      err[b] = std::sqrt(s / (double)ctx->env_elems);  // diffrax rms_norm
    }
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

}  // namespace

int tsit5_trial(pdeopt_ctx* ctx, double t, double dt, double rtol, double atol, double* err) {
  return ctx->prob.dtype == PDEOPT_F32 ? tsit5_trial_t<float>(ctx, t, dt, rtol, atol, err)
                                       : tsit5_trial_t<double>(ctx, t, dt, rtol, atol, err);
}

template <typename T>
static int tsit5_dense_t(pdeopt_ctx* ctx, double theta, double dt, int env_first, int env_count, void* dev_out) {
  double b[7];
  tsit5_dense_weights(theta, b);
  const int64_t o = (int64_t)env_first * (int64_t)ctx->env_elems;
  LinComb<T> lc{};
  lc.y = (const T*)ctx->Y + o;
  lc.n = 7;
  for (int j = 0; j < 7; ++j) {
    lc.k[j] = (const T*)ctx->K[j] + o;
    lc.c[j] = T(dt * b[j]);
  }
  lc.out = (T*)dev_out + o;
  const int64_t total = (int64_t)ctx->env_elems * env_count;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(lincomb_kernel<T>, dim3(blocks), dim3(256), 0, ctx->stream, lc, total);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

int tsit5_dense(pdeopt_ctx* ctx, double theta, double dt, int env_first, int env_count, void* dev_out) {
  if (!ctx->tsit5_pending) return fail(ctx, PDEOPT_ESTATE, "dense output needs a pending Tsit5 trial step");
  return ctx->prob.dtype == PDEOPT_F32 ? tsit5_dense_t<float>(ctx, theta, dt, env_first, env_count, dev_out)
                                       : tsit5_dense_t<double>(ctx, theta, dt, env_first, env_count, dev_out);
}

// FSAL slope of environment b *= ratio[b] (its step size changed between two per-environment trials)
int tsit5_rescale_fsal(pdeopt_ctx* ctx, const double* ratio) {
  const size_t need = sizeof(double) * (size_t)ctx->prob.batch;
  if (ctx->red_cap < need) {
    if (ctx->red_dev) (void)hipFree(ctx->red_dev);
    ctx->red_dev = nullptr;
    ctx->red_cap = 0;
    PDEOPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->red_dev, need));
    ctx->red_cap = need;
  }
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->red_dev, ratio, need, hipMemcpyHostToDevice, ctx->stream));
  const int64_t ee = (int64_t)ctx->env_elems;
  const dim3 grid((unsigned)std::min<int64_t>((ee + 255) / 256, 1024), ctx->prob.batch);
  if (ctx->prob.dtype == PDEOPT_F32)
    hipLaunchKernelGGL(env_scale_kernel<float>, grid, dim3(256), 0, ctx->stream, (float*)ctx->K[0], (const double*)ctx->red_dev, ee);
  else
    hipLaunchKernelGGL(env_scale_kernel<double>, grid, dim3(256), 0, ctx->stream, (double*)ctx->K[0], (const double*)ctx->red_dev, ee);
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // `ratio` is the caller's again
  return PDEOPT_OK;
}

// per-environment commit: accepted environments take the candidate and its FSAL slope, the others keep theirs
int tsit5_commit_env(pdeopt_ctx* ctx, const uint8_t* accept) {
  if (!ctx->tsit5_pending) return fail(ctx, PDEOPT_ESTATE, "no Tsit5 trial step is pending");
  ctx->tsit5_pending = false;
  const int batch = ctx->prob.batch;
  std::swap(ctx->Y, ctx->TB);
  std::swap(ctx->K[0], ctx->K[6]);
  const size_t eb = ctx->env_elems * ctx->esize;
  for (int b = 0; b < batch; ++b) {
    if (accept[b]) continue;
    // rejected: the old state / slope (now in TB / K[6]) go back
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync((char*)ctx->Y + eb * b, (const char*)ctx->TB + eb * b, eb, hipMemcpyDeviceToDevice, ctx->stream));
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync((char*)ctx->K[0] + eb * b, (const char*)ctx->K[6] + eb * b, eb, hipMemcpyDeviceToDevice, ctx->stream));
  }
  ctx->tsit5_fsal_valid = true;
  return PDEOPT_OK;
}

int tsit5_commit(pdeopt_ctx* ctx, int accept) {
  if (!ctx->tsit5_pending) return fail(ctx, PDEOPT_ESTATE, "no Tsit5 trial step is pending");
  ctx->tsit5_pending = false;
  if (accept) {
    std::swap(ctx->Y, ctx->TB);         // y <- 5th-order candidate
    std::swap(ctx->K[0], ctx->K[6]);    // FSAL: k7 = f(t+dt, y1) is the next k1
  }
  ctx->tsit5_fsal_valid = true;         // on rejection K[0] = f(t, y) is still valid
  return PDEOPT_OK;
}

// the in-kernel adaptive solve: one workgroup per environment (LDS-resident periodic Cahn-Hilliard / Allen-Cahn grids,
// stencil_small_adaptive.hpp) or several cooperating workgroups per environment (larger grids, the smoothed-boundary
// forms, advection-diffusion: stencil_coop_adaptive.hpp)
static bool tsit5_single_wg(const pdeopt_ctx* ctx) {
  return ctx->prob.dtype == PDEOPT_F32 ? small_tsit5_supported<float>(ctx) : small_tsit5_supported<double>(ctx);
}
static bool tsit5_coop(const pdeopt_ctx* ctx) {
  return ctx->prob.dtype == PDEOPT_F32 ? coop_tsit5_supported<float>(ctx) : coop_tsit5_supported<double>(ctx);
}
bool tsit5_solve_small_supported(const pdeopt_ctx* ctx) { return tsit5_single_wg(ctx) || tsit5_coop(ctx); }

int tsit5_solve_small(pdeopt_ctx* ctx, double t0, double t1, double dt0, const pdeopt_pid* pid, int64_t max_steps, int n_save,
                      const double* save_ts, void* save_host, pdeopt_tsit5_stats* stats) {
  const bool single = tsit5_single_wg(ctx), coop = tsit5_coop(ctx);
  if (!single && !coop)
    return fail(ctx, PDEOPT_EINVAL, "the in-kernel adaptive solve takes periodic / smoothed-boundary Cahn-Hilliard and Allen-Cahn FD problems and "
                                    "advection-diffusion with a steady velocity, on grids its tiles cover (pdeopt_tsit5_solve_small_supported)");
  // both apply: one workgroup up to two vectors per thread (64^2 fp32: 18 us per trial step against 21 on 32 workgroups);
  // from three vectors per thread on (its state then lives in LDS: 64^2 fp64 33 us) the multi-workgroup kernel (25 us)
  constexpr int kV32 = 4, kV64 = 2;
  const int64_t nvec = (int64_t)ctx->prob.nx * ctx->prob.ny / (ctx->prob.dtype == PDEOPT_F32 ? kV32 : kV64);
  const bool prefer_coop = ctx->opt_small_persist == 2 || (ctx->opt_small_persist == 0 && nvec > 1024);
  if (coop && (!single || prefer_coop))
    return ctx->prob.dtype == PDEOPT_F32 ? coop_tsit5_solve<float>(ctx, t0, t1, dt0, pid, max_steps, n_save, save_ts, save_host, stats)
                                         : coop_tsit5_solve<double>(ctx, t0, t1, dt0, pid, max_steps, n_save, save_ts, save_host, stats);
  return ctx->prob.dtype == PDEOPT_F32 ? small_tsit5_solve<float>(ctx, t0, t1, dt0, pid, max_steps, n_save, save_ts, save_host, stats)
                                       : small_tsit5_solve<double>(ctx, t0, t1, dt0, pid, max_steps, n_save, save_ts, save_host, stats);
}

int launch_lerp(pdeopt_ctx* ctx, const void* a, const void* b, void* out, double theta,
                size_t env_first, size_t env_count) {
  const size_t off = env_first * ctx->env_elems;
  const int64_t n = (int64_t)(env_count * ctx->env_elems);
  const int threads = 256;
  const int blocks = (int)std::min<int64_t>((n + threads - 1) / threads, 2048);
  if (ctx->prob.dtype == PDEOPT_F32) {
    hipLaunchKernelGGL(lerp_kernel<float>, dim3(blocks), dim3(threads), 0, ctx->stream,
                       (const float*)a + off, (const float*)b + off, (float*)out, (float)theta, n);
  } else {
    hipLaunchKernelGGL(lerp_kernel<double>, dim3(blocks), dim3(threads), 0, ctx->stream,
                       (const double*)a + off, (const double*)b + off, (double*)out, theta, n);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  return PDEOPT_OK;
}

}  // namespace pdeopt
