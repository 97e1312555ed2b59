// Internal declarations shared by the translation units of libpdeopt_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pdeopt_hip.h"

namespace pdeopt {

constexpr int kMaxCoef = PDEOPT_CLOSURE_MAX_COEF;
constexpr int kNumAux = 8;

// Per-environment scalar parameters, stored in the arithmetic type of the path.  One struct per
// environment in device memory; kernels index it with the (wave-uniform) batch index so the
// loads are scalar.
template <typename T>
struct EnvParams {
  T kappa;
  T gpe_k;
  T kscale;  // slope scale dt_b / dt_ref of per-environment step sizes (pdeopt_tsit5_trial_env); 1 otherwise
  T imex_scale;  // sigma_b: this environment's fourier_symbol = sigma_b x the uploaded one (pdeopt_set_env_imex_scale)
  T mu[kMaxCoef];
  T mob[kMaxCoef];
  T fe[kMaxCoef];  // free-energy density closure (smoothed-boundary equations)
};

// Gaussian light spots of the GPE control field (pdeopt_set_gpe_spots), in the arithmetic type of the path;
// device table [batch][PDEOPT_MAX_SPOTS]
template <typename T>
struct LightSpot {
  T amp0, amp_rate, x0, x_rate, y0, y_rate, c;  // c = 1 / (2 width^2)
};
// what a Strang kernel needs to evaluate lights(t, x, y) of its environment at cell (ix, iy)
template <typename T>
struct SpotArgs {
  const LightSpot<T>* table;  // nullptr / n == 0: no spots
  int n;
  T t, x_first, y_first, hx, hy;
};
template <typename T>
__device__ __forceinline__ T t_exp_neg(T x);  // exp(-x)
template <>
__device__ __forceinline__ float t_exp_neg<float>(float x) { return __expf(-x); }
template <>
__device__ __forceinline__ double t_exp_neg<double>(double x) { return exp(-x); }
template <typename T>
__device__ __forceinline__ T spots_value(const SpotArgs<T>& a, int env, T x, T y) {
  T w = T(0);
  for (int s = 0; s < a.n; ++s) {
    const LightSpot<T> q = a.table[(size_t)env * PDEOPT_MAX_SPOTS + s];
    const T dx = x - (q.x0 + q.x_rate * a.t), dy = y - (q.y0 + q.y_rate * a.t);
    w += (q.amp0 + q.amp_rate * a.t) * t_exp_neg<T>((dx * dx + dy * dy) * q.c);
  }
  return w;
}

// structure of a closure (shared by the whole batch; only coefficient VALUES vary per env)
struct ClosureSpec {
  int kind;
  int flags;
  int n;
};

// Where a field lives in memory.  Periodic fields: ld == ny, off == 0, wrap by index.
// Halo-padded tiles (domain decomposition): neighbours exist in memory, no wrap.
struct Geo {
  int nx, ny;       // extent of the computed region
  int ld;           // row pitch in elements
  int64_t off;      // element offset of cell (0,0) inside one environment's array
  int64_t bstride;  // elements between environments
  int periodic;     // 1: wrap indices; 0: read halo cells at i in [-2, nx+1], j in [-2, ny+1]
  int nz;           // 3-D equations: extent of the contiguous axis (1 otherwise)
};

struct AuxField {
  void* dev = nullptr;
  int per_env = 0;
  size_t bytes = 0;
  // time-dependent source (pdeopt_set_aux_time_fn): host callback, pinned staging buffer, time last uploaded
  pdeopt_aux_fn fn = nullptr;
  void* user = nullptr;
  void* stage = nullptr;
  double t_loaded = 0.0;
  bool loaded = false;
};

struct CommState;    // RCCL / in-process communicator + strip buffers of the decomposed driver (comm.hip)
struct Spectral;     // rocFFT plans + work buffers (spectral.hip)
struct StrangFused;  // LDS-FFT split-step state (strang_fused.hip)

// what a captured substep graph depends on (explicit integrators, stencil.hip)
struct GraphStructure {
  int equation, dtype, nx, ny, nz, batch, derivs;
  double hx, hy, hz;
  int mu_kind, mu_flags, mu_n, mob_kind, mob_flags, mob_n;
};
struct GraphKey {
  int integrator;
  int fused;
  double dt;
  void *Y, *TA, *TB, *ACC, *KS, *ep, *vx, *vy;
  int64_t kernel_path, tile_rows, ablate;
  GraphStructure structure;
};

}  // namespace pdeopt

struct pdeopt_ctx {
  int device = 0;
  int num_cus = 256;  // compute units of the device (sizes persistent launches)
  hipStream_t stream = nullptr;
  bool stream_borrowed = false;  // stream belongs to the caller (pdeopt_ctx_create_on_stream)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  void* clock_stamps = nullptr;   // [start, stop] x (s_memtime, s_memrealtime) of the timer pair
  double timer_shader_hz = 0.0;   // shader clock held between the last pdeopt_timer_start / _stop (0: not measured)
  std::string err;
  std::string last_kernel;
  bool configured = false;
  pdeopt_problem prob{};
  size_t esize = 4;        // bytes per real element
  int comps = 1;           // 2 for the GPE (re, im)
  size_t env_elems = 0;    // real elements per environment
  size_t total_bytes = 0;  // one field, whole batch
  // field buffers (device).  Y always holds the current state.
  void* Y = nullptr;
  void* TA = nullptr;
  void* TB = nullptr;
  void* ACC = nullptr;
  void* SNAP = nullptr;
  void* obs_dev = nullptr;  // uint8 observation frames
  void* vort_dev = nullptr; // vortex counters + winding map
  size_t vort_cap = 0;
  void* KS = nullptr;  // slope scratch of the spectral-RHS stage path
  void* K[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // Tsit5 slopes
  bool tsit5_pending = false;
  bool tsit5_fsal_valid = false;
  bool slope_scaled = false;           // stage launches multiply k by EnvParams::kscale (per-environment dt)
  std::vector<double> kscale_prev;     // scale the FSAL slope K[0] of each environment carries
  // smoothed-boundary equations: time of the RHS evaluation being launched and its scalar terms
  double cur_t = 0.0;
  pdeopt_time_fn time_fn = nullptr;
  void* time_user = nullptr;
  double time_const[3] = {0.0, 0.0, 0.0};
  // pdeopt_set_time_terms_poly: theta(t), flux(t) of the smoothed-boundary forms as cubic polynomials in t -- what the
  // in-kernel adaptive solve (stencil_coop_adaptive.hpp) evaluates at its own stage times
  bool time_poly_valid = false;
  double time_theta[4] = {0.0, 0.0, 0.0, 0.0}, time_flux[4] = {0.0, 0.0, 0.0, 0.0};
  // pdeopt_set_time_table: the three scalars at a list of evaluation times, looked up before the callback is asked
  std::vector<double> tt_times, tt_terms;
  size_t tt_cursor = 0;
  int64_t tt_misses = 0;  // stage times a non-empty table did not hold while no callback was registered
  void* env_params_dev = nullptr;
  std::vector<char> env_params_host;
  pdeopt::AuxField aux[pdeopt::kNumAux];
  std::string jit_src[2];  // pdeopt_set_jit_closures: C function bodies of mu / mobility (closure kind PDEOPT_CL_JIT)
  int64_t opt_kernel_path = 0;
  int64_t opt_tile_rows = 0;  // 0 auto, 16 or 32
  int64_t opt_group_envs = 0; // explicit integrators: envs per cache-resident group (0 auto, <0 whole batch)
  int64_t n_stage_launches = 0;  // fused stencil+update launches issued so far
  int64_t opt_halo = 0;          // requested halo width of the NEXT configure (0 periodic, 4 or 8 padded)
  int halo = 0;                  // halo width of the configured layout
  int pair_ext = 0;              // next PAIR_12 launch covers the tile + this many ring cells (halo-8 layout: 4)
  void* pair_strip = nullptr;    // next PAIR_34 launch also writes the tile's halo strip here (fused pack)
  const void* pair_recv = nullptr;  // next PAIR_12 launch reads the halo from these gathered strips (fused unpack)
  int pair_nbr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const void* pair_peer[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // ... or from the neighbours' own strip buffers (peer-mapped exchange)
  void* halo_scratch = nullptr;  // single-rank loop-back buffer for pack/unpack
  void* halo_scratch2 = nullptr; // second loop-back strip (halo-8 loop: fused pack writes one while the next is read)
  size_t halo_scratch_bytes = 0, halo_scratch2_bytes = 0;
  int64_t opt_imex_lds_fft = 0;  // IMEX transforms: 0 auto (hand-written passes where the size is covered), -1 rocFFT
  int64_t opt_graph = 0;         // hipGraph replay of the substep loop: 0 auto (launch-bound sizes), 1 always, -1 never
  hipGraphExec_t graph_exec = nullptr;
  pdeopt::GraphKey graph_key{};
  int64_t graph_launches_per_replay = 0;
  std::string graph_name;
  int64_t opt_fuse_stages = 0;   // RK4 stage-pair fusion: 0 auto, -1 off (one launch per stage), 1 stage pairs (AC: no single-pass kernel)
  int64_t opt_debug_ablate = 0;  // timing-only ablations, results are wrong when set
  void* adaptive_blk = nullptr;   // save times + statistics + save slots of pdeopt_tsit5_solve_small
  size_t adaptive_cap = 0;
  int64_t opt_group_streams = 0;  // PDEOPT_OPT_GROUP_STREAMS
  hipStream_t stream2 = nullptr;  // second stream of the two-groups-side-by-side schedule (created on first use)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int64_t opt_small_persist = 0; // whole-environment-step kernel for LDS-resident grids: 0 auto, 1 wherever it can, -1 never
  // Gaussian light spots of the GPE (pdeopt_set_gpe_spots)
  int n_spots = 0;
  void* spots_dev = nullptr;
  std::vector<pdeopt_light_spot> spots_host;  // [batch][PDEOPT_MAX_SPOTS]
  double spots_x_first = 0.0, spots_y_first = 0.0;
  bool imex_per_env = false;  // some environment has imex_scale != 1: one environment per complex field
  int launch_part = 0;        // tiles of the next stage-pair launches: 0 all, 1 interior, 2 edge (pdeopt_rk4_phase_part)
  int win_lo = 0, win_n = 0;  // environment window the stage launchers operate on
  int64_t last_group_streams = 1;  // PDEOPT_CNT_GROUP_STREAMS
  int64_t last_groups = 1;    // environment groups of the last advance (PDEOPT_CNT_LAST_GROUPS)
  double imex_A = 0.5, ts_re = 1.0, ts_im = 0.0, strang_dx = 1.0;
  double* red_dev = nullptr;  // reduction scratch
  size_t red_cap = 0;
  double* red_mean_dev = nullptr;
  std::vector<void*> host_allocs;  // pdeopt_host_alloc
  pdeopt::CommState* comm = nullptr;
  pdeopt::Spectral* spectral = nullptr;
  pdeopt::StrangFused* strang_fused = nullptr;
};

namespace pdeopt {

// Padded (domain-decomposition) layouts.  halo 4: the tile with 4 halo cells on every side.  halo 8: 8 halo cells
// in front of the tile and 8 + a tail margin behind it -- the first stage pair of a substep runs on the tile + 4
// ring in whole workgroup tiles (32 rows x 32 16-byte vectors) that start at cell (-4, -4), so its last tile row /
// column over-runs the ring by up to a tile; the over-run cells are computed from whatever the margin holds and
// written back into it, nobody reads them.  Rows: 8 + nx + 8 + 32; row pitch: 8 + ny + 8 + 128.
constexpr int kTailRows = 32, kTailCols = 128;
inline int pad_rows(int nx, int h) { return h == 8 ? nx + 2 * h + kTailRows : nx + 2 * h; }
inline int pad_ld(int ny, int h) { return h == 8 ? ny + 2 * h + kTailCols : ny + 2 * h; }

// field geometry of the configured layout: periodic, or padded by ctx->halo cells on every side
inline Geo make_geo(const pdeopt_ctx* ctx) {
  Geo g;
  const int h = ctx->halo;
  g.nx = ctx->prob.nx;
  g.ny = ctx->prob.ny;
  g.ld = pad_ld(ctx->prob.ny, h);
  g.off = (int64_t)h * g.ld + h;
  g.bstride = (int64_t)pad_rows(ctx->prob.nx, h) * g.ld;
  g.periodic = h == 0 ? 1 : 0;
  g.nz = ctx->prob.nz > 1 ? ctx->prob.nz : 1;
  if (g.nz > 1) g.bstride *= g.nz;  // [nx][ny][nz], no halo layout in 3-D
  return g;
}

int fail(pdeopt_ctx* ctx, int code, const char* fmt, ...);

#define PDEOPT_HIP_CHECK(ctx, expr)                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return pdeopt::fail((ctx), PDEOPT_EHIP, "%s failed: %s (%s:%d)", #expr,              \
                          hipGetErrorString(e_), __FILE__, __LINE__);                      \
  } while (0)

int ensure_buffer(pdeopt_ctx* ctx, void** p, size_t bytes);
int ensure_stream2(pdeopt_ctx* ctx);  // the ctx's second stream + fork / join events, created on first use
// api.hip: bring a time-dependent auxiliary field to local time t (no-op for static fields)
int refresh_time_aux(pdeopt_ctx* ctx, int which, double t);
inline bool has_time_aux(const pdeopt_ctx* ctx, int which) { return ctx->aux[which].fn != nullptr; }
// where the next fused-unpack launch finds the strips of its 8 neighbours: slices of the gathered buffer (pair_recv +
// rank * strip elements), or the neighbours' own buffers (pair_peer, peer-mapped exchange); all nullptr: no fused unpack
template <typename T>
inline void fill_neighbour_strips(const pdeopt_ctx* ctx, int64_t strip_rank_elems, const T** nbase) {
  for (int q = 0; q < 8; ++q) {
    if (ctx->pair_peer[0]) nbase[q] = static_cast<const T*>(ctx->pair_peer[q]);
    else if (ctx->pair_recv) nbase[q] = static_cast<const T*>(ctx->pair_recv) + (int64_t)ctx->pair_nbr[q] * strip_rank_elems;
    else nbase[q] = nullptr;
  }
}
// light spots of the environments from win_lo on, at local time t
template <typename T>
inline SpotArgs<T> make_spot_args(const pdeopt_ctx* ctx, double t) {
  SpotArgs<T> a{};
  a.n = ctx->n_spots;
  a.table = a.n ? static_cast<const LightSpot<T>*>(ctx->spots_dev) + (size_t)ctx->win_lo * PDEOPT_MAX_SPOTS : nullptr;
  a.t = T(t);
  a.x_first = T(ctx->spots_x_first);
  a.y_first = T(ctx->spots_y_first);
  a.hx = T(ctx->prob.hx);
  a.hy = T(ctx->prob.hy);
  return a;
}

// stencil.hip
int launch_rhs(pdeopt_ctx* ctx, const void* in, void* out, double t);
// the same slope for the IMEX step: the stage-B half of the fused pair kernel where it applies (CH, periodic
// layout, polynomial / logit closures), launch_rhs otherwise; equal to launch_rhs up to rounding
int launch_rhs_slope(pdeopt_ctx* ctx, const void* in, void* out, double t);
int advance_explicit(pdeopt_ctx* ctx, int integrator, double t0, double dt, int64_t n);
int tsit5_trial(pdeopt_ctx* ctx, double t, double dt, double rtol, double atol, double* err);
int tsit5_commit(pdeopt_ctx* ctx, int accept);
int tsit5_dense(pdeopt_ctx* ctx, double theta, double dt, int env_first, int env_count, void* dev_out);
int tsit5_commit_env(pdeopt_ctx* ctx, const uint8_t* accept);
int tsit5_rescale_fsal(pdeopt_ctx* ctx, const double* ratio);  // K[0] of environment b *= ratio[b]
bool tsit5_solve_small_supported(const pdeopt_ctx* ctx);
int tsit5_solve_small(pdeopt_ctx* ctx, double t0, double t1, double dt0, const pdeopt_pid* pid, int64_t max_steps, int n_save,
                      const double* save_ts, void* save_host, pdeopt_tsit5_stats* stats);
void graph_destroy(pdeopt_ctx* ctx);
int launch_lerp(pdeopt_ctx* ctx, const void* a, const void* b, void* out, double theta,
                size_t env_first, size_t env_count);
// halo.hip (domain decomposition: padded, non-periodic layout)
int halo_pack(pdeopt_ctx* ctx, int field, void* dev_send);
int halo_unpack(pdeopt_ctx* ctx, int field, const void* dev_recv, const int* nbr);
size_t halo_strip_elems(const pdeopt_ctx* ctx);
int rk4_phase(pdeopt_ctx* ctx, int phase, double dt, int part = 0);
int rk4_loopback_advance(pdeopt_ctx* ctx, double dt, int64_t n);
// halo-8 layout: one substep = [unpack(Y), or fused into the first pair] -> PAIR_12 on the tile + 4 ring -> PAIR_34 (+ the new state's halo strip
// written into `strip` by the edge tiles' epilogue when strip != nullptr); Y / TA are swapped
// recv != nullptr: the halo of the state is NOT in the field yet -- the first pair's edge tiles read it from the
// gathered strips `recv` (neighbour ranks nbr[8]) and write the frame cells back (fused unpack)
int rk4_substep_h8(pdeopt_ctx* ctx, double dt, void* strip, const void* recv = nullptr, const int* nbr = nullptr);
int rk4_substep_h8_peer(pdeopt_ctx* ctx, double dt, void* strip, const void* const* peer);
int rk4_phase_plan(pdeopt_ctx* ctx, int* fields, int* nphases);
void* field_ptr(pdeopt_ctx* ctx, int field);
// comm.hip
int comm_unique_id(pdeopt_ctx* ctx, char* out128);
int comm_init(pdeopt_ctx* ctx, int world, int rank, const char* id128);
int comm_init_local(pdeopt_ctx* ctx, pdeopt_local_group* g, int rank);
pdeopt_local_group* local_group_new(int world);
void local_group_delete(pdeopt_local_group* g);
void comm_destroy(pdeopt_ctx* ctx);
int rk4_decomposed_advance(pdeopt_ctx* ctx, double dt, int64_t n, const int* nbr, int overlap);
int comm_ipc_export(pdeopt_ctx* ctx, int world, int rank, void* handle64);
int comm_ipc_attach(pdeopt_ctx* ctx, const void* handles);
// reduce.hip
int reduce_state(pdeopt_ctx* ctx, int op, double* out);
int probe_state(pdeopt_ctx* ctx, const int32_t* cells, int n_probes, int env_first, int env_count, double* host_out);
int observe_u8(pdeopt_ctx* ctx, double lo, double hi, int env_first, int env_count, void* host_out);
int detect_vortices(pdeopt_ctx* ctx, double amp_thresh, double tol, int env_first, int env_count,
                    int32_t* host_winding, int64_t* host_counts);
// spectral.hip
int advance_imex(pdeopt_ctx* ctx, double t0, double dt, int64_t n);
int rhs_fourier(pdeopt_ctx* ctx, const void* in, void* out);
int advance_strang(pdeopt_ctx* ctx, double t0, double dt, int64_t n);
void spectral_destroy(pdeopt_ctx* ctx);
void spectral_invalidate(pdeopt_ctx* ctx);
// jit.hip: do these closure bodies compile (hiprtc, gfx950)?
int jit_check(int dtype, const char* mu_body, const char* mob_body, char* log_out, int log_cap);
// strang_fused.hip
bool strang_fused_supported(const pdeopt_ctx* ctx);
int advance_strang_fused(pdeopt_ctx* ctx, double t0, double dt, int64_t n);
bool imex_fused_supported(const pdeopt_ctx* ctx);
int advance_imex_fused(pdeopt_ctx* ctx, double dt, int64_t n);
void strang_fused_invalidate(pdeopt_ctx* ctx);
void strang_fused_destroy(pdeopt_ctx* ctx);

}  // namespace pdeopt
