/*
 * pdeopt_hip.h -- C ABI of libpdeopt_hip.so, the MI355X (gfx950) hot path of pde_opt.
 *
 * The reference (acoh64/pde-opt @ 2025-09-26) has no FFI: its "plugin boundary" is the
 * Python class surface.  The hot path it runs per environment step is
 *
 *     PDEEnv.step            pde_opt/pde_env.py:244-317      (diffeqsolve at :293-303)
 *     PDEModel.solve         pde_opt/pde_model.py:68-136     (diffeqsolve at :120-134)
 *       -> solver.step       pde_opt/numerics/solvers.py:56-70, :99-122 (+ diffrax Euler/ERK)
 *         -> equation.rhs    pde_opt/numerics/equations/{cahn_hilliard,allen_cahn,gross_pitaevskii}.py
 *           -> FD stencils   pde_opt/numerics/utils/derivatives.py:8-61, jnp.fft
 *
 * Everything below "diffeqsolve(...)" is replaced by the entry points declared here; the
 * Python shims in pde_opt_amd/ (same class names and signatures as the reference) call
 * them through ctypes.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every call returns a pdeopt_status (0 = ok); pdeopt_last_error() gives the text.
 *   - caller owns host memory; the library owns device memory (hipMalloc on the ctx device).
 *   - one ctx per (thread, device); calls on a ctx are ordered on the ctx's HIP stream and are
 *     not re-entrant.  pdeopt_advance / pdeopt_rhs are asynchronous; get_state / reduce /
 *     sync / timer_stop synchronise.
 *   - fields are C-order [batch][nx][ny] (axis 1 contiguous), exactly the reference's
 *     (Nx, Ny) arrays stacked over a leading batch axis; the GPE state is [batch][nx][ny][2]
 *     (re, im interleaved == the reference's (N, N, 2) layout, gross_pitaevskii.py:75).
 */
#ifndef PDEOPT_HIP_H
#define PDEOPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pdeopt_ctx pdeopt_ctx;

typedef enum {
  PDEOPT_OK = 0,
  PDEOPT_EINVAL = 1,       /* bad argument / unsupported combination (reference: ValueError) */
  PDEOPT_EHIP = 2,         /* HIP runtime error */
  PDEOPT_EFFT = 3,         /* rocFFT error */
  PDEOPT_ENONFINITE = 4,   /* reported by pdeopt_reduce(op = NONFINITE) callers */
  PDEOPT_ENOMEM = 5,
  PDEOPT_ESTATE = 6        /* call order error (e.g. advance before configure) */
} pdeopt_status;

typedef enum { PDEOPT_F32 = 0, PDEOPT_F64 = 1 } pdeopt_dtype;

/* which equation.rhs() the stencil kernel evaluates */
typedef enum {
  PDEOPT_EQ_CAHN_HILLIARD = 0,   /* cahn_hilliard.py:89-109  div(D(u) grad(mu_h(u) - kappa lap u)) */
  PDEOPT_EQ_ALLEN_CAHN = 1,      /* allen_cahn.py:81-84      -R(u) (mu_h(u) - kappa lap u)          */
  PDEOPT_EQ_ADVECTION_DIFFUSION = 2, /* not in the reference package: -div(v u) + D lap u */
  PDEOPT_EQ_GPE = 3,             /* gross_pitaevskii.py:67-75 (Strang only)                         */
  /* smoothed-boundary variants: psi = domain.geometry.smooth is an aux field */
  PDEOPT_EQ_ALLEN_CAHN_SBM = 4,     /* allen_cahn.py:88-159    -R (mu_h - kappa/psi div(psi grad u) - wall sqrt(2 f)) */
  PDEOPT_EQ_CAHN_HILLIARD_SBM = 5,  /* cahn_hilliard.py:204-289 div(psi D grad(inner))/psi + source                  */
  PDEOPT_EQ_CAHN_HILLIARD_3D = 6,   /* cahn_hilliard.py:113-200: fields are [batch][nx][ny][nz] */
  /* shapes.py:39-64, Shape.smooth_shape: u_t = 2 (c lap u + (1 - c) u_nn) - 18 u (1 - u)(1 - 2u) / eps^2 with centred
   * differences, u_nn the second derivative along grad u.  pdeopt_problem.kappa carries c (smooth_curvature),
   * pdeopt_problem.gpe_k carries eps (smooth_epsilon); no closures */
  PDEOPT_EQ_SHAPE_SMOOTH = 7
} pdeopt_equation;

/* which solver.step() is fused around it */
typedef enum {
  PDEOPT_INT_EULER = 0,   /* diffrax.Euler  (pde_env.py:293-303 call site)                 */
  PDEOPT_INT_RK4 = 1,     /* classical RK4 (new; BASELINE.json configs 2,3,5)              */
  PDEOPT_INT_IMEX = 2,    /* SemiImplicitFourierSpectral.step   numerics/solvers.py:56-70  */
  PDEOPT_INT_STRANG = 3,  /* StrangSplitting.step               numerics/solvers.py:99-122 */
  PDEOPT_INT_TSIT5 = 4    /* diffrax.Tsit5 fixed step (adaptive PID driven from the host)  */
} pdeopt_integrator;

/* Pointwise closure family standing in for the reference's Python callables mu(u), D(u), R(u)
 * (dataclass fields cahn_hilliard.py:51-54, allen_cahn.py:47-50; catalogue: SURVEY Appendix D).
 *     s(c)  = sum_k coef[k] * c^k                      (kind POLY)
 *           = sum_k coef[k] * P_k(2c - 1)              (kind LEGENDRE, legendre.py:23-34,50,70)
 *     f(c)  = s(c) [+ log(c / (1 - c)) if LOGIT_PRIOR] [+ c log c + (1-c) log(1-c) if MIX_ENTROPY]
 *             then  exp(.) if EXP_WRAP                                                        */
#define PDEOPT_CLOSURE_MAX_COEF 16
typedef enum {
  PDEOPT_CL_POLY = 0,
  PDEOPT_CL_LEGENDRE = 1,
  PDEOPT_CL_JIT = 2 /* a callable outside the family: its C function body was handed over with pdeopt_set_jit_closures and
                       is compiled at run time (hiprtc) into the generic Cahn-Hilliard / Allen-Cahn stencil kernel; no
                       coefficients (constants are part of the body).  2-D periodic / padded FD problems only. */
} pdeopt_closure_kind;
#define PDEOPT_CL_LOGIT_PRIOR 1
#define PDEOPT_CL_EXP_WRAP 2
#define PDEOPT_CL_MIX_ENTROPY 4  /* + c log c + (1 - c) log(1 - c): ideal mixing entropy of the regular-
                                    solution free energy f(c) (notebooks/smooth_boundary.ipynb)          */

typedef struct {
  int32_t kind;   /* pdeopt_closure_kind */
  int32_t flags;  /* PDEOPT_CL_* bits */
  int32_t n;      /* number of coefficients, 1..PDEOPT_CLOSURE_MAX_COEF */
  int32_t reserved;
  double coef[PDEOPT_CLOSURE_MAX_COEF];
} pdeopt_closure;

typedef enum {
  PDEOPT_DERIVS_FD = 0,      /* rhs_fd:      cahn_hilliard.py:89-109, allen_cahn.py:81-84 (stencil kernels) */
  PDEOPT_DERIVS_FOURIER = 1  /* rhs_fourier: cahn_hilliard.py:82-87,  allen_cahn.py:74-79 (rocFFT)          */
} pdeopt_derivs;

typedef struct {
  int32_t equation;   /* pdeopt_equation */
  int32_t dtype;      /* pdeopt_dtype: arithmetic type of the whole path */
  int32_t nx, ny;     /* Domain.points (domains.py:24) */
  int32_t batch;      /* independent environments advanced in lock step (new; >= 1) */
  int32_t derivs;     /* pdeopt_derivs: "fd" | "fourier" switch of the equations (cahn_hilliard.py:75-80) */
  double hx, hy;      /* Domain.dx (domains.py:30-33) */
  double kappa;       /* gradient-energy coefficient (CH/AC); diffusion coefficient D (AD) */
  pdeopt_closure mu;  /* mu_h (CH/AC) */
  pdeopt_closure mob; /* D (CH) or R (AC) */
  /* GPE (gross_pitaevskii.py:35-44): b = -i (V + k |psi|^2); V is an auxiliary field */
  double gpe_k;
  pdeopt_closure fe;  /* free-energy density f(u) of the smoothed-boundary equations (allen_cahn.py:112) */
  int32_t nz;         /* third extent (contiguous axis) of the 3-D equations; 0 or 1 for 2-D problems */
  int32_t reserved2;
  double hz;
} pdeopt_problem;

/* auxiliary read-only fields (pdeopt_set_aux) */
typedef enum {
  PDEOPT_AUX_VX_FACE = 0,    /* AD: x velocity on faces (i+1/2, j), real [nx][ny]                    */
  PDEOPT_AUX_VY_FACE = 1,    /* AD: y velocity on faces (i, j+1/2), real [nx][ny]                    */
  PDEOPT_AUX_IMEX_SYMBOL = 2,/* IMEX: fourier_symbol, complex [nx][ny] (cahn_hilliard.py:74)         */
  PDEOPT_AUX_GPE_A_TERM = 3, /* Strang: A_term, complex [nx][ny] (gross_pitaevskii.py:62)            */
  PDEOPT_AUX_GPE_POTENTIAL = 4,/* Strang: V = 1/2 tf((1+e)X^2+(1-e)Y^2) + lights(t,X,Y), real [nx][ny] */
  PDEOPT_AUX_SBM_PSI = 5,      /* SBM: level-set field psi = domain.geometry.smooth, real [nx][ny]            */
  PDEOPT_AUX_SBM_NORM_GRAD = 6,/* SBM: norm_grad_psi = |grad_c psi| / psi, real [nx][ny] (allen_cahn.py:128-133) */
  PDEOPT_AUX_SBM_MASK = 7      /* SBM: left_half, the 0/1 field selecting which wall carries cos(theta)
                                  (allen_cahn.py:134-135, cahn_hilliard.py:253-254), real [nx][ny]            */
} pdeopt_aux;

/* Time-dependent scalars of the smoothed-boundary equations, evaluated by the host at every RHS
 * evaluation time t (each Runge-Kutta stage has its own):
 *   out[0] = cos(theta(t))                       weight of the wall term where MASK = 1
 *   out[1] = cos(pi - theta(t)) (CH) or 0 (AC)   weight of the wall term where MASK = 0
 *   out[2] = flux(t)  (CH; 0 for AC)             normal boundary flux J_n
 * (allen_cahn.py:150-154, cahn_hilliard.py:269-275,289).  The kernel forms
 *   wall = sqrt(kappa) NORM_GRAD (out[0] MASK + out[1] (1 - MASK)),  source = NORM_GRAD out[2]. */
typedef void (*pdeopt_time_fn)(double t, double out[3], void* user);

typedef enum {
  PDEOPT_RED_MEAN = 0,
  PDEOPT_RED_VAR = 1,       /* population variance == np.var (notebooks/test_pde_env.ipynb:57) */
  PDEOPT_RED_MIN = 2,
  PDEOPT_RED_MAX = 3,
  PDEOPT_RED_SUMSQ = 4,     /* sum |.|^2 (GPE norm without the dx^2 factor) */
  PDEOPT_RED_NONFINITE = 5  /* count of NaN/Inf cells */
} pdeopt_reduce_op;

/* kernel-selection knobs (pdeopt_set_option) */
typedef enum {
  PDEOPT_OPT_KERNEL_PATH = 0, /* 0 = auto, 1 = force the generic (untiled) kernels, 2 = force LDS-tiled */
  PDEOPT_OPT_TILE_ROWS = 1,   /* rows per LDS tile: 0 = auto, 16 or 32; 64: the 64 x 64 tile of the single-pass
                                 Cahn-Hilliard kernel */
  PDEOPT_OPT_GROUP_ENVS = 2,  /* explicit integrators: advance the batch in groups of this many
                                 environments so a group's working set stays in the 256 MiB
                                 Infinity Cache across stages and substeps (0 = auto, < 0 = whole
                                 batch in one sweep) */
  PDEOPT_OPT_IMEX_LDS_FFT = 7,/* IMEX transforms: 0 = auto (the hand-written register/LDS FFT passes, two real
                                 environments per complex field, for power-of-two sizes 64..1024; rocFFT
                                 real<->hermitian plans otherwise), -1 = always rocFFT */
  PDEOPT_OPT_GRAPH = 6,       /* explicit integrators: replay the substep loop from a captured hipGraph:
                                 0 = auto (launch-bound problem sizes), 1 = always, -1 = never */
  PDEOPT_OPT_HALO_LAYOUT = 5, /* layout of the NEXT pdeopt_configure: 0 = periodic field (wrap by index),
                                 4 = rank-local tile padded by a 4-cell halo on every side, no wrap
                                 (domain decomposition; halos filled by pdeopt_halo_unpack; one exchange per
                                 RK4 phase: 2 per substep with fused stage pairs, 4 otherwise),
                                 8 = 8-cell halo (+ a tail margin): ONE exchange per RK4 substep -- the first
                                 stage pair is evaluated on the tile + 4 ring, so the second finds its input there
                                 (fused Cahn-Hilliard stage pairs only; pdeopt_rk4_phase_plan reports {0, -1}) */
  PDEOPT_OPT_FUSE_STAGES = 4, /* RK4: temporally fused stages: 0 = auto (the whole substep in one pass over HBM where
                                 such a kernel exists: fp32 Allen-Cahn; fp32 Cahn-Hilliard on periodic grids of 32 x 128
                                 tiles; stage pairs 1+2 / 3+4 otherwise), 1 = stage pairs only, 2 = as 0,
                                 -1 = off (one launch per stage) */
  PDEOPT_OPT_SMALL_PERSIST = 8,/* Euler / RK4 on grids whose stage input (+ chemical potential) fits one compute unit's
                                 LDS (CH <= 128^2 fp32, 96^2 fp64): ALL n substeps of pdeopt_advance in ONE launch, one
                                 workgroup per environment, the state in registers (the sizes of the reference's own
                                 tests and notebooks, where a launch per stage pair is latency-bound): 0 = auto (grids
                                 up to 4096 cells, larger ones from 192 environments on), 1 = wherever it can run,
                                 -1 = never; 2 = as 1, and the multi-workgroup kernel (stencil_coop_adaptive.hpp: several
                                 compute units per environment) is preferred wherever it can run -- for the adaptive
                                 solve and for Euler / RK4.  In auto mode that kernel takes Euler / RK4 advances of >= 8
                                 substeps of up to 16 environments of 65^2 ... 192^2 cells (fp64: from 64^2 on) and the adaptive solves one compute unit cannot hold */
  PDEOPT_OPT_GROUP_STREAMS = 9,/* explicit integrators running the batch in cache-resident groups: 0 = auto (two groups
                                  side by side on two HIP streams, each half the size, so that one group's launch
                                  fills the other's ramp and tail), 1 = one group at a time, 2 = force two */
  PDEOPT_OPT_DEBUG_ABLATE = 3 /* TIMING ONLY, results are wrong: bit0 skip the mu phase, bit1 skip
                                 the flux phase of the tiled kernel (where does the time go?) */
} pdeopt_option;

/* ---- life cycle ------------------------------------------------------------------------- */
int pdeopt_abi_version(void);
int pdeopt_device_count(int* count);
int pdeopt_ctx_create(int device, pdeopt_ctx** out);
int pdeopt_ctx_destroy(pdeopt_ctx* ctx);
const char* pdeopt_last_error(const pdeopt_ctx* ctx); /* ctx may be NULL: last create() error */
int pdeopt_set_option(pdeopt_ctx* ctx, int option, int64_t value);

/* ---- problem set-up == equation_type(domain=..., **parameters)  (pde_env.py:286) -------- */
int pdeopt_configure(pdeopt_ctx* ctx, const pdeopt_problem* problem);
/* per-environment overrides of kappa and closure coefficients (control parameters travel with
 * the environment).  Any pointer may be NULL (= leave unchanged).  mu_coef / mob_coef are
 * [env_count][PDEOPT_CLOSURE_MAX_COEF]. */
int pdeopt_set_env_params(pdeopt_ctx* ctx, int env_first, int env_count, const double* kappa,
                          const double* mu_coef, const double* mob_coef);
/* IMEX with one implicit operator PER ENVIRONMENT: environment b integrates with fourier_symbol_b = sigma[b] x the
 * uploaded IMEX_SYMBOL (cahn_hilliard.py:74: fourier_symbol = kappa (2 pi i k)^4, so sigma[b] = kappa_b / kappa_ref
 * when kappa is the per-environment control).  Needs a real symbol.  While some sigma != 1 the hand-written FFT
 * passes carry ONE environment per complex field instead of two (a pair shares its multiplier), the multiplier is
 * formed in the column pass as 1 / (N (1 + sigma_b A dt symbol)); all sigma == 1 restores the paired passes. */
int pdeopt_set_env_imex_scale(pdeopt_ctx* ctx, int env_first, int env_count, const double* sigma);
/* per-environment GPE interaction strength k (gross_pitaevskii.py:38-39): with a batch every
 * environment carries its own control value (BASELINE config 4 is an RL environment whose agent may act on k) */
int pdeopt_set_env_gpe_k(pdeopt_ctx* ctx, int env_first, int env_count, const double* k);
/* shared (per_env = 0: [nx][ny]) or per-environment (per_env = 1: [batch][nx][ny]) auxiliary
 * field, host pointer, element type = problem dtype (complex = 2 elements).  Replaces a time-dependent
 * source registered with pdeopt_set_aux_time_fn for the same field. */
int pdeopt_set_aux(pdeopt_ctx* ctx, int which, const void* host, int per_env);
/* Time-dependent auxiliary field.  The reference evaluates its callables at the time of every right-hand
 * side:  Strang  b = terms.vf(t0, y0, args) with control(t) = lights(t, X, Y) at each substep's t0
 * (numerics/solvers.py:109, gross_pitaevskii.py:61,67-75); explicit integrators at every stage time.
 * fn(t, which, host_out, user) fills host_out ([nx][ny], or [batch][nx][ny] with per_env = 1, problem
 * dtype) with the field at local time t and returns 0 (non-zero aborts the call with PDEOPT_EINVAL).
 * It is called on the calling thread from inside pdeopt_advance / pdeopt_rhs / pdeopt_tsit5_trial, once
 * per distinct evaluation time, for GPE_POTENTIAL (once per Strang substep) and VX_FACE / VY_FACE (once
 * per stage); the substep loop then runs the whole batch in one group and without hipGraph replay.
 * fn = NULL removes the source (the field keeps its last contents). */
typedef int (*pdeopt_aux_fn)(double t, int which, void* host_out, void* user);
int pdeopt_set_aux_time_fn(pdeopt_ctx* ctx, int which, pdeopt_aux_fn fn, void* user, int per_env);
/* The GPE control field lights(t, X, Y) (gross_pitaevskii.py:43,61,72) as a sum of Gaussian spots that the
 * Strang kernels evaluate themselves at every substep's t0 -- the stirring-beam controls of the RL environments
 * without a host round trip per substep:
 *     lights(t, x, y) = sum_s (amp0 + amp_rate t) exp(-((x - x0 - x_rate t)^2 + (y - y0 - y_rate t)^2) inv_two_w2)
 * added to the GPE_POTENTIAL aux field (which then holds the time-independent part: the trap).  spots is
 * [env_count][n_spots]; every environment of a batch carries its own spots (n_spots is one number per ctx,
 * 0 .. PDEOPT_MAX_SPOTS; 0 removes them).  x_first / y_first: coordinates of cell (0, 0) (Domain.axes()[d][0],
 * domains.py:36-40); the spacing is the problem's hx, hy. */
#define PDEOPT_MAX_SPOTS 4
typedef struct {
  double amp0, amp_rate;
  double x0, x_rate;
  double y0, y_rate;
  double inv_two_w2; /* 1 / (2 width^2) */
} pdeopt_light_spot;
int pdeopt_set_gpe_spots(pdeopt_ctx* ctx, int env_first, int env_count, int n_spots, const pdeopt_light_spot* spots,
                         double x_first, double y_first);

/* ---- state  == PDEEnv._state (pde_env.py:232,305) ---------------------------------------- */
int pdeopt_set_state(pdeopt_ctx* ctx, int env_first, int env_count, const void* host);
int pdeopt_get_state(pdeopt_ctx* ctx, int env_first, int env_count, void* host);
/* device pointer of the current state (valid until the next advance); for zero-copy consumers */
int pdeopt_state_device_ptr(pdeopt_ctx* ctx, void** dev_ptr, int64_t* bytes);

/* ---- compute ------------------------------------------------------------------------------ */
/* out = equation.rhs(state, t) for every environment; host_out may be NULL (compute only). */
int pdeopt_rhs(pdeopt_ctx* ctx, double t, void* host_out);
/* n_substeps of size dt starting at local time t0:  the body of diffeqsolve's while-loop
 * under ConstantStepSize.  Asynchronous -- except on the multi-workgroup path (a few mid-sized environments, see
 * PDEOPT_OPT_SMALL_PERSIST), which returns once its one launch has finished and reports a launch whose workgroups
 * could not all become resident as an error instead of hanging. */
int pdeopt_advance(pdeopt_ctx* ctx, int integrator, double t0, double dt, int64_t n_substeps);
/* Closures outside the in-kernel family (the reference accepts any pointwise callable: cahn_hilliard.py:51-54,
 * allen_cahn.py:47-50, functions/legendre.py:56-74 prior_fn): mu_body / mob_body are C function BODIES -- one line of
 * statements ending in `return <expression>;`, the argument is `c`, the scalar type is `T`, literals are written T(1.5),
 * available: + - * / exp log tanh sqrt pow jit_powi(x, n) -- for the closures whose kind is PDEOPT_CL_JIT in the NEXT
 * pdeopt_configure (NULL for a role that stays in the family).  The host mirror emits them from the traced sympy
 * expression of the callable (a vetted node set, never user text).  Compiled once per distinct (bodies, dtype) on first
 * use; a body that does not compile makes that call fail with the compiler's message. */
int pdeopt_set_jit_closures(pdeopt_ctx* ctx, const char* mu_body, const char* mob_body);
/* do these two bodies compile (hiprtc, for gfx950; no device is needed)?  0 = yes; the compiler's log (warnings, or the
 * errors) is copied into log[0 .. log_cap).  What pdeopt_configure + the first launch would find out on the GPU box. */
int pdeopt_jit_check(int dtype, const char* mu_body, const char* mob_body, char* log, int log_cap);
/* Smoothed-boundary equations: fn is called on the calling thread, once per RHS evaluation, from
 * inside pdeopt_rhs / pdeopt_advance / pdeopt_tsit5_trial; fn == NULL uses constant[3] instead.  */
int pdeopt_set_time_terms(pdeopt_ctx* ctx, pdeopt_time_fn fn, void* user, const double constant[3]);
/* The same scalars for a LIST of evaluation times, handed over before pdeopt_advance: terms is [n][3] = out[0..2] of
 * pdeopt_time_fn at times[0..n-1].  A right-hand side evaluated at a listed time (compared exactly: form the stage
 * times as the library does -- t0 + s dt, + dt/2, + dt for RK4) takes its scalars from the table, so a fixed-step
 * advance of n substeps makes no host callback at all (one per stage otherwise: 400 Python calls per 100 RK4
 * substeps); other times still go to fn / constant.  n = 0 clears it; pdeopt_set_time_terms clears it too. */
int pdeopt_set_time_table(pdeopt_ctx* ctx, int n, const double* times, const double* terms);
/* The same two callables as POLYNOMIALS in t -- theta(t) = sum theta[i] t^i, flux(t) likewise, at most cubic -- for the
 * code that chooses its own evaluation times on the device: the in-kernel adaptive solve (pdeopt_tsit5_solve_small)
 * forms cos(theta(t)), cos(pi - theta(t)), flux(t) at every stage time itself.  Call it AFTER pdeopt_set_time_terms
 * (which withdraws earlier polynomials); the callback / constants still serve every other path, so they must describe
 * the same functions.  n_theta = 0 withdraws the polynomials (a callback without them keeps the adaptive solve on the
 * host-driven path).  Reference: theta / flux fields of cahn_hilliard.py:232-235, allen_cahn.py:119-120;
 * notebooks/smooth_boundary.ipynb:262 (a quadratic theta(t)). */
int pdeopt_set_time_terms_poly(pdeopt_ctx* ctx, int n_theta, const double* theta, int n_flux, const double* flux);
/* integrator parameters: IMEX A (solvers.py:43); Strang time_scale re/im and dx (solvers.py:86-89) */
int pdeopt_set_integrator_params(pdeopt_ctx* ctx, double imex_A, double time_scale_re,
                                 double time_scale_im, double strang_dx);
/* snapshot <- state ;   host_out <- snapshot + theta (state - snapshot): LocalLinearInterpolation
 * dense output for SaveAt(ts=...) (solvers.py:48,91; pde_model.py:128) */
int pdeopt_snapshot(pdeopt_ctx* ctx);
int pdeopt_get_interpolated(pdeopt_ctx* ctx, double theta, int env_first, int env_count,
                            void* host_out);
/* per-environment reductions of the state (reward helpers); out is [batch] doubles. */
int pdeopt_reduce(pdeopt_ctx* ctx, int op, double* out_per_env);
/* Point probes: the state at n_probes grid cells -- (i, j) index pairs, (i, j, k) triples for the 3-D
 * equations -- of every environment of a range; host_out is [env_count][n_probes][comps] doubles (comps = 2
 * for the GPE: re, im).  Sensor-style observations / rewards without moving the field. */
int pdeopt_probe(pdeopt_ctx* ctx, const int32_t* cells, int n_probes, int env_first, int env_count,
                 double* host_out);
/* uint8 image observations formed on the device (observation space Box(0, 255, (1, *points), uint8),
 * pde_env.py:118-126): q = rint(clip((x - lo) / (hi - lo), 0, 1) * 255); host_out is
 * [env_count][nx][ny] bytes -- a quarter (fp32) / an eighth (fp64) of the D2H of the raw field. */
int pdeopt_observe_u8(pdeopt_ctx* ctx, double lo, double hi, int env_first, int env_count,
                      uint8_t* host_out);
/* The same frames left in device memory for a consumer on the same GPU (an RL policy in PyTorch-ROCm): *dev_out
 * is a library-owned buffer of env_count * nx * ny bytes, valid until the next pdeopt_observe_u8* call or
 * pdeopt_configure with another shape; the call returns after the ctx stream has produced it, so any stream may
 * read it.  No reference counterpart: upstream hands observations to the host (pde_env.py:305-312). */
int pdeopt_observe_u8_device(pdeopt_ctx* ctx, double lo, double hi, int env_first, int env_count, void** dev_out,
                             int64_t* nbytes);
/* rl_utils.detect_vortices (pde_opt/rl_utils.py:19-84) on the resident GPE wavefunction: integer phase
 * winding of every grid plaquette, |circulation| < tol * 2 pi and cells whose corner-averaged density
 * is below amp_thresh (if > 0) suppressed.  host_counts is [env_count][3] = {num_vortices,
 * total_topological_charge, abs_charge_count}; host_winding ([env_count][nx][ny] int32) may be NULL
 * when only the counts are wanted (24 bytes per environment instead of the field). */
int pdeopt_detect_vortices(pdeopt_ctx* ctx, double amp_thresh, double tol, int env_first, int env_count,
                           int32_t* host_winding, int64_t* host_counts);

/* ---- domain decomposition of one large field (BASELINE config 5; no reference counterpart) -----
 * With PDEOPT_OPT_HALO_LAYOUT = 4 a ctx holds one rank's tile.  Per RK4 substep the caller runs
 *     for phase in 0..nphases-1:  pack(fields[phase]) -> all-gather strips -> unpack -> rk4_phase(phase)
 * fields[] / nphases come from pdeopt_rk4_phase_plan (2 phases with fused stage pairs, else 4).
 * Field ids: 0 = state Y, 1 = TA, 2 = TB, 3 = ACC.  A strip is pdeopt_halo_strip_elems() elements of
 * the problem dtype: [top h rows][bottom h rows][left h cols][right h cols][TL][TR][BL][BR] of the
 * tile INTERIOR.  dev_send / dev_recv are DEVICE pointers (e.g. torch tensors handed to RCCL);
 * dev_recv holds the strips of all ranks, rank-major.  neighbours[8] = ranks of
 * {up, down, left, right, up-left, up-right, down-left, down-right} (up = smaller x index).
 * NULL dev_send / dev_recv selects an internal loop-back buffer (single rank, every neighbour 0).
 * Halo layout 8: h = 8 in the strip layout, and pdeopt_rk4_phase_plan reports nphases = 2 with fields = {0, -1}:
 * only phase 0 is preceded by an exchange (of the state), phase 1 finds the ring of its input computed by phase 0. */
int pdeopt_halo_strip_elems(pdeopt_ctx* ctx, int64_t* elems);
int pdeopt_halo_pack(pdeopt_ctx* ctx, int field, void* dev_send);
int pdeopt_halo_unpack(pdeopt_ctx* ctx, int field, const void* dev_recv, const int* neighbours);
int pdeopt_rk4_phase_plan(pdeopt_ctx* ctx, int* fields /* [4] */, int* nphases);
int pdeopt_rk4_phase(pdeopt_ctx* ctx, int phase, double dt);
/* The same phase split in two launches so that the halo exchange can be in flight while most of the tile is
 * computed: part 1 = the INTERIOR workgroup tiles (they read no halo cell), part 2 = the EDGE tiles (first / last
 * tile row and column), part 0 = everything (== pdeopt_rk4_phase).  Per phase:
 *     pack(field) -> [all-gather on a second stream]  ||  rk4_phase_part(phase, dt, 1)
 *     -> unpack(field) -> rk4_phase_part(phase, dt, 2)
 * Fused stage pairs only (pdeopt_rk4_phase_plan reports 2 phases); the substep's buffer rotation happens with
 * part 2 (or 0) of the last phase. */
typedef enum { PDEOPT_PART_ALL = 0, PDEOPT_PART_INTERIOR = 1, PDEOPT_PART_EDGE = 2 } pdeopt_tile_part;
int pdeopt_rk4_phase_part(pdeopt_ctx* ctx, int phase, double dt, int part);
/* n RK4 substeps of a SINGLE-RANK padded tile, loop-back halo exchange included, in one call (every neighbour
 * is the tile itself: the periodic problem in the decomposed layout; one rank's share of the decomposed
 * driver without the collective). */
int pdeopt_rk4_loopback_advance(pdeopt_ctx* ctx, double dt, int64_t n_substeps);
/* The whole decomposed substep loop inside the library, on an RCCL communicator of its own: one process per
 * GPU, the host language only carries rank 0's 128-byte id to the other ranks (pdeopt_comm_unique_id ->
 * broadcast by any means -> pdeopt_comm_init on every rank's ctx).  pdeopt_rk4_decomposed_advance then runs
 * n substeps of   pack -> ncclAllGather of strips -> unpack -> phase   with no host round trip per substep;
 * overlap != 0 puts the collective on a second HIP stream and computes the interior tiles of the phase while
 * it is in flight (fused stage pairs).  RCCL is resolved at run time (dlopen of the librccl.so the process
 * already uses), the library has no link-time dependency on it. */
int pdeopt_comm_unique_id(char out[128]);
int pdeopt_comm_init(pdeopt_ctx* ctx, int world, int rank, const char id[128]);
int pdeopt_comm_destroy(pdeopt_ctx* ctx);
/* The same loop on an IN-PROCESS group of ranks instead of RCCL: the ranks are ctxs of one process -- several on one
 * GPU ("virtual ranks": every N > 1 line of the decomposed driver runs without a second GPU) or one per GPU with a
 * host thread each -- and the all-gather is device-to-device copies between their strip buffers, ordered by HIP
 * events.  Every rank's thread must be inside pdeopt_rk4_decomposed_advance at the same time (they rendezvous once
 * per exchange; a rank that does not show up within 60 s fails the call on every rank).  Destroy the members'
 * communicators (pdeopt_comm_destroy / pdeopt_ctx_destroy) before the group. */
typedef struct pdeopt_local_group pdeopt_local_group;
int pdeopt_local_group_create(int world, pdeopt_local_group** out);
int pdeopt_local_group_destroy(pdeopt_local_group* group);
int pdeopt_comm_init_local(pdeopt_ctx* ctx, pdeopt_local_group* group, int rank);
/* The same loop with NO collective ("P2P stores into peer-mapped halo buffers"): one process per GPU as
 * under RCCL; every rank allocates its two strip buffers + three counters once (pdeopt_comm_ipc_export, after the tile
 * has been configured in the halo-8 layout: returns the 64-byte hipIpc handle of the block), the host language carries
 * the handles between the processes (any transport), and pdeopt_comm_ipc_attach(all world x 64 bytes, rank-major) maps
 * every other rank's block.  In pdeopt_rk4_decomposed_advance a rank's stencil kernel then reads its 8 neighbours'
 * strips IN PLACE through the mapped pointers and writes its own next strip; two monotone counters per rank (published
 * by a one-lane kernel behind the stencil kernel, polled by a wait kernel in front of it) order the exchanges.  A
 * neighbour that does not publish within 2 s fails the call instead of hanging.  Processes sharing one GPU work too
 * (that is how the test-suite runs it). */
int pdeopt_comm_ipc_export(pdeopt_ctx* ctx, int world, int rank, void* handle64);
int pdeopt_comm_ipc_attach(pdeopt_ctx* ctx, const void* handles /* [world][64] */);
int pdeopt_rk4_decomposed_advance(pdeopt_ctx* ctx, double dt, int64_t n_substeps, const int* neighbours /* [8] */,
                                  int overlap);
/* ctx whose work is ordered on a caller-owned HIP stream (e.g. torch's current stream, so RCCL
 * collectives issued through torch.distributed order with the kernels without host syncs) */
int pdeopt_ctx_create_on_stream(int device, void* hip_stream, pdeopt_ctx** out);

/* caller-owned device buffers (halo strips when no tensor library is at hand) */
typedef enum { PDEOPT_COPY_H2D = 0, PDEOPT_COPY_D2H = 1, PDEOPT_COPY_D2D = 2 } pdeopt_copy_kind;
/* page-locked HOST memory for results fetched every environment step (uint8 observation frames, states): a
 * D2H copy into it runs at the full PCIe rate and without the page faults a freshly allocated pageable buffer
 * takes (32 MiB of frames: ~0.7 ms instead of ~4.5 ms).  Freed by pdeopt_host_free or with the ctx. */
int pdeopt_host_alloc(pdeopt_ctx* ctx, int64_t bytes, void** host);
int pdeopt_host_free(pdeopt_ctx* ctx, void* host);
int pdeopt_buffer_alloc(pdeopt_ctx* ctx, int64_t bytes, void** dev);
int pdeopt_buffer_free(pdeopt_ctx* ctx, void* dev);
int pdeopt_buffer_copy(pdeopt_ctx* ctx, void* dst, const void* src, int64_t bytes, int kind);

/* ---- Tsit5 building blocks for host-driven adaptive stepping (row f1) ----------------------- */
/* one trial step of size dt from the current state: on return err_norm[batch] holds the
 * diffrax-style scaled RMS error norm per environment (atol + rtol max(|y0|,|y1|)).  The trial
 * result is kept pending until pdeopt_tsit5_commit(accept). */
int pdeopt_tsit5_trial(pdeopt_ctx* ctx, double t, double dt, double rtol, double atol,
                       double* err_norm);
int pdeopt_tsit5_commit(pdeopt_ctx* ctx, int accept);
/* The same with ONE STEP SIZE PER ENVIRONMENT ("per-env error-norm reduction and per-env dt"):
 * environment b steps by dt[b] (0 = leave it alone); the stages still run as one batched launch, every slope
 * is stored scaled by dt[b] / dt_ref with dt_ref = max_b dt[b] (returned), so the shared coefficients
 * dt_ref a_ij apply to all environments.  Needs an autonomous right-hand side (environments sit at different
 * times).  pdeopt_tsit5_commit_env accepts / rejects per environment (accept[batch], 0 or 1);
 * pdeopt_tsit5_dense takes dt = dt_ref and theta of the environment asked for. */
int pdeopt_tsit5_trial_env(pdeopt_ctx* ctx, double t, const double* dt, double rtol, double atol, double* dt_ref,
                           double* err_norm);
int pdeopt_tsit5_commit_env(pdeopt_ctx* ctx, const uint8_t* accept);
/* Dense output of the PENDING trial step (call between pdeopt_tsit5_trial and pdeopt_tsit5_commit):
 * host_out = y(t + theta dt) = y + dt sum_i b_i(theta) k_i, theta in [0, 1], with the 4th-order continuous
 * extension of Tsitouras' pair (Comput. Math. Appl. 62 (2011), section 4) -- the interpolant diffrax.Tsit5
 * evaluates for SaveAt(ts=...) points that fall inside a step (tests/test_solvers.py:86-96 saves 200 of them). */
int pdeopt_tsit5_dense(pdeopt_ctx* ctx, double theta, double dt, int env_first, int env_count, void* host_out);

/* The whole adaptive solve t0 -> t1 in ONE launch, controller included, for grids that fit a compute unit's LDS
 * (Cahn-Hilliard / Allen-Cahn, FD, analytic closure classes; up to 2048 vectors of 16 bytes per environment: 64 x 128
 * in fp32, 64 x 64 in fp64).  Replaces, for those grids, the host loop the reference runs through
 * diffrax.diffeqsolve(..., Tsit5(), stepsize_controller=PIDController(rtol, atol, ...)) (pde_opt/pde_model.py:100-118;
 * tests/test_solvers.py:81,263): trial step, scaled RMS error norm, PID factor, accept / reject and dense output all
 * happen inside the kernel, one workgroup per environment.  EVERY ENVIRONMENT RUNS ITS OWN CONTROLLER -- what batch == 1
 * and PIDController(per_environment=True) mean; a step size shared by several environments is the caller's loop over
 * pdeopt_tsit5_trial / pdeopt_tsit5_commit.
 *   pid        diffrax.PIDController's fields; dtmin = -HUGE_VAL / dtmax = HUGE_VAL for "none"
 *   max_steps  trial steps (accepted + rejected) per environment; an environment that reaches it stops where it is
 *              (status PDEOPT_TSIT5_MAX_STEPS), as does one whose step can no longer advance t (PDEOPT_TSIT5_STALLED)
 *   save_ts    n_save ascending times in (t0, t1]; host_save [n_save][batch][nx][ny] receives the 4th-order dense
 *              output there (slots an environment never reached are NaN)
 *   stats      [batch]
 * Returns PDEOPT_EINVAL, leaving the state untouched, when the configured problem is not one the kernel takes
 * (ask pdeopt_tsit5_solve_small_supported first). */
typedef struct {
  double rtol, atol, pcoeff, icoeff, dcoeff, dtmin, dtmax, factormin, factormax, safety;
} pdeopt_pid;
typedef enum { PDEOPT_TSIT5_DONE = 0, PDEOPT_TSIT5_MAX_STEPS = 1, PDEOPT_TSIT5_STALLED = 2 } pdeopt_tsit5_status;
typedef struct {
  double t, dt;               /* where the environment stopped; the step size the controller would try next */
  int64_t accepted, rejected; /* trial steps */
  int32_t status;             /* pdeopt_tsit5_status */
  int32_t saved;              /* save points written */
} pdeopt_tsit5_stats;
int pdeopt_tsit5_solve_small_supported(pdeopt_ctx* ctx);
int pdeopt_tsit5_solve_small(pdeopt_ctx* ctx, double t0, double t1, double dt0, const pdeopt_pid* pid, int64_t max_steps,
                             int n_save, const double* save_ts, void* host_save, pdeopt_tsit5_stats* stats);

/* ---- timing / sync ------------------------------------------------------------------------- */
int pdeopt_sync(pdeopt_ctx* ctx);
int pdeopt_timer_start(pdeopt_ctx* ctx);           /* hipEventRecord on the ctx stream */
int pdeopt_timer_stop(pdeopt_ctx* ctx, double* ms); /* record + synchronise + elapsed */
/* the shader clock the chip HELD between the last timer_start / timer_stop pair, in Hz: both calls also stamp
 * s_memtime (shader-clock ticks) and s_memrealtime (constant 100 MHz) on the ctx stream; 0 if not measured.
 * (bench.py prices VALU cycles on this instead of a data-sheet clock.) */
int pdeopt_timer_clock(pdeopt_ctx* ctx, double* shader_hz);
typedef enum {
  PDEOPT_CNT_STAGE_LAUNCHES = 0, /* kernel launches of the integrators so far: fused stencil + update launches, and the
                                    FFT passes of the hand-written Strang / IMEX pipelines */
  PDEOPT_CNT_LAST_GROUPS = 1,    /* environment groups the last pdeopt_advance ran the batch in (1 = one sweep) */
  PDEOPT_CNT_GROUP_STREAMS = 2   /* groups the last pdeopt_advance kept in flight side by side (PDEOPT_OPT_GROUP_STREAMS): 1 or 2 */
} pdeopt_counter;
int pdeopt_get_counter(pdeopt_ctx* ctx, int which, int64_t* value);
/* name of the kernel variant the last advance/rhs dispatched (for tests and profiles) */
const char* pdeopt_last_kernel(const pdeopt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PDEOPT_HIP_H */
