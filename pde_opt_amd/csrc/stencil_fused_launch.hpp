// Host-side dispatch of the fused stage-pair kernels (included after every kernel header so that the
// kernel templates are defined before their launch sites are instantiated).
#pragma once

#include "stencil_fused.hpp"
#include "stencil_fused_ac.hpp"

namespace pdeopt {

template <typename T>
int launch_pair(pdeopt_ctx* ctx, int pair, const void* in, const void* y, const void* acc, void* out,
                void* acc_out, double aA, double bA, double aB, double bB) {
  const pdeopt_problem& p = ctx->prob;
  PairArgs<T> s{};
  s.g = make_geo(ctx);
  const int64_t woff = (int64_t)ctx->win_lo * s.g.bstride;
  s.in = static_cast<const T*>(in) + woff;
  s.y = y ? static_cast<const T*>(y) + woff : nullptr;
  s.acc = acc ? static_cast<const T*>(acc) + woff : nullptr;
  s.out = static_cast<T*>(out) + woff;
  s.acc_out = acc_out ? static_cast<T*>(acc_out) + woff : nullptr;
  s.aA = T(aA); s.bA = T(bA); s.aB = T(aB); s.bB = T(bB);
  s.rhx = T(0.5 / (p.hx * p.hx)); s.rhy = T(0.5 / (p.hy * p.hy));
  s.rhx2 = T(1.0 / (p.hx * p.hx)); s.rhy2 = T(1.0 / (p.hy * p.hy));
  s.ep = static_cast<const EnvParams<T>*>(ctx->env_params_dev) + ctx->win_lo;
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  s.dbg = (int)ctx->opt_debug_ablate;
  s.part = ctx->launch_part;
  // halo-8 layout of a decomposed field: PAIR_12 on the tile + 4 ring, PAIR_34 with the fused pack
  int ext = 0;
  if (ctx->halo == 8) {
    if (p.equation != PDEOPT_EQ_CAHN_HILLIARD) return fail(ctx, PDEOPT_EINVAL, "the halo-8 layout runs the fused Cahn-Hilliard stage pairs only");
    if (pair == PAIR_12) {
      ext = ctx->pair_ext;
      const int64_t shift = (int64_t)ext * s.g.ld + ext;
      s.in -= shift; s.out -= shift; s.acc_out -= shift;
      s.ext = ext;
      // fused unpack: halo cells straight from the neighbours' strips
      s.strip_env = 2LL * 8 * p.ny + 2LL * p.nx * 8 + 4LL * 64;
      fill_neighbour_strips<T>(ctx, s.strip_env * p.batch, s.nbase);
    } else if (ctx->pair_strip) {
      s.strip = static_cast<T*>(ctx->pair_strip);
      s.strip_env = 2LL * 8 * p.ny + 2LL * p.nx * 8 + 4LL * 64;
    }
  }
  ctx->n_stage_launches++;
  const int cl = classify_closures(p.mu, p.mob);
  // tile height of the pair kernels.  CH: 32 rows (512-thread workgroups) where they divide the grid --
  // less redundant ring work at the same VGPR count, +4.5 % on 1024^2 same-box once the kernel sat at
  // 78 / 88 VGPRs -- otherwise 16; PDEOPT_OPT_TILE_ROWS overrides.  AC follows the per-stage kernels.
  int rpt = tiled_rpt(ctx);
  if (p.equation == PDEOPT_EQ_CAHN_HILLIARD && ctx->opt_tile_rows == 0) rpt = (p.nx % 32 == 0) ? 4 : 2;
  char name[96];
  snprintf(name, sizeof(name), "stage_pair<%s,%s,%s,rows%d>", sizeof(T) == 4 ? "f32" : "f64",
           p.equation == PDEOPT_EQ_ALLEN_CAHN ? "AC" : "CH", cl == CL_LOGIT ? "logit" : "poly", 8 * rpt);
  ctx->last_kernel = name;
  if (cl == CL_LOGIT && p.mu.n <= 2 && p.equation == PDEOPT_EQ_CAHN_HILLIARD) {
    // linear polynomial part: the shorter closure (same bits, see closures.hpp)
    if (rpt == 2)
      return pair == PAIR_12 ? launch_pair_ch_inst<T, CL_LOGIT1, PAIR_12, 2>(ctx, s, ext)
                             : launch_pair_ch_inst<T, CL_LOGIT1, PAIR_34, 2>(ctx, s, 0);
    return pair == PAIR_12 ? launch_pair_ch_inst<T, CL_LOGIT1, PAIR_12, 4>(ctx, s, ext)
                           : launch_pair_ch_inst<T, CL_LOGIT1, PAIR_34, 4>(ctx, s, 0);
  }
#define PDEOPT_PAIR_DISPATCH(CLV, PAIRV)                                             \
  (rpt == 2 ? launch_pair_inst<T, CLV, PAIRV, 2>(ctx, s, ext) : launch_pair_inst<T, CLV, PAIRV, 4>(ctx, s, ext))
  if (cl == CL_LOGIT)
    return pair == PAIR_12 ? PDEOPT_PAIR_DISPATCH(CL_LOGIT, PAIR_12) : PDEOPT_PAIR_DISPATCH(CL_LOGIT, PAIR_34);
  return pair == PAIR_12 ? PDEOPT_PAIR_DISPATCH(CL_POLY, PAIR_12) : PDEOPT_PAIR_DISPATCH(CL_POLY, PAIR_34);
#undef PDEOPT_PAIR_DISPATCH
}

// out = f(in) for every environment of the window through the stage-B half of the fused kernel
// (PAIR_K): the slope launch of the IMEX step.  Same arithmetic as launch_pair's stages (including the
// folded mu form of the linear-logit class), so it may differ from the per-stage kernels' k by rounding;
// pdeopt_rhs and the explicit integrators do not come here.
template <typename T>
bool slope_pair_supported(const pdeopt_ctx* ctx) {
  return ctx->prob.equation == PDEOPT_EQ_CAHN_HILLIARD && !ctx->halo && ctx->opt_fuse_stages >= 0 &&
         ctx->opt_kernel_path != 1 && fused_supported<T>(ctx);
}
template <typename T>
int launch_slope_pair(pdeopt_ctx* ctx, const void* in, void* out) {
  const pdeopt_problem& p = ctx->prob;
  PairArgs<T> s{};
  s.g = make_geo(ctx);
  const int64_t woff = (int64_t)ctx->win_lo * s.g.bstride;
  s.in = static_cast<const T*>(in) + woff;
  s.out = static_cast<T*>(out) + woff;
  s.rhx = T(0.5 / (p.hx * p.hx)); s.rhy = T(0.5 / (p.hy * p.hy));
  s.rhx2 = T(1.0 / (p.hx * p.hx)); s.rhy2 = T(1.0 / (p.hy * p.hy));
  s.ep = static_cast<const EnvParams<T>*>(ctx->env_params_dev) + ctx->win_lo;
  s.mu = ClosureSpec{p.mu.kind, p.mu.flags, p.mu.n};
  s.mob = ClosureSpec{p.mob.kind, p.mob.flags, p.mob.n};
  ctx->n_stage_launches++;
  const int cl = classify_closures(p.mu, p.mob);
  const bool rows32 = ctx->opt_tile_rows == 32 || (ctx->opt_tile_rows == 0 && p.nx % 32 == 0);
  ctx->last_kernel = rows32 ? "slope_pair<CH,rows32>" : "slope_pair<CH,rows16>";
#define PDEOPT_SLOPE(CLV) \
  (rows32 ? launch_pair_ch_inst<T, CLV, PAIR_K, 4>(ctx, s, 0) : launch_pair_ch_inst<T, CLV, PAIR_K, 2>(ctx, s, 0))
  if (cl == CL_LOGIT) return p.mu.n <= 2 ? PDEOPT_SLOPE(CL_LOGIT1) : PDEOPT_SLOPE(CL_LOGIT);
  return PDEOPT_SLOPE(CL_POLY);
#undef PDEOPT_SLOPE
}

}  // namespace pdeopt
