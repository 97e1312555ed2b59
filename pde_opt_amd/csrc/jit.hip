// Run-time-compiled closures ("generic escape hatch = hiprtc from a vetted expression"; VERDICT r3
// missing #4).  The reference takes any pointwise callable as mu / D / R (cahn_hilliard.py:51-54, allen_cahn.py:47-50;
// functions/legendre.py:56-74 `prior_fn: Callable`); callables outside the in-kernel family (closures.hpp) arrive here as
// C function bodies emitted by the host from the traced sympy expression (pde_opt_amd/numerics/closures.py: jit_body,
// a vetted node set: + * pow exp log tanh sqrt and rational constants) -- pdeopt_set_jit_closures -- and the generic
// stage kernel of jit_device.hpp is compiled with them by hiprtc, once per distinct (bodies, dtype), cached for the
// life of the process.  Only when the family match fails: the fast paths never come here.
//
// hiprtc is resolved at run time (dlopen "libhiprtc.so"): the library carries no link-time dependency on it and a
// caller who never passes such a closure never loads it.
#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <map>
#include <mutex>

#include "stencil_generic.hpp"

namespace pdeopt {

namespace {

const char kJitSource[] =
#include "build/jit_device_source.inc"
    ;

struct Rtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
};

struct JitModule {
  hipModule_t mod = nullptr;
  hipFunction_t ch = nullptr, ac = nullptr;
};

std::mutex g_mu;
Rtc g_rtc;
std::map<std::string, JitModule> g_modules;  // key: device + dtype + bodies

int load_rtc(pdeopt_ctx* ctx) {
  if (g_rtc.lib) return PDEOPT_OK;
  for (const char* n : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})
    if (!g_rtc.lib) g_rtc.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!g_rtc.lib) return fail(ctx, PDEOPT_EINVAL, "closures outside the in-kernel family are compiled at run time: cannot load libhiprtc.so (%s)", dlerror());
#define PDEOPT_RTC_SYM(field, name) g_rtc.field = reinterpret_cast<decltype(g_rtc.field)>(dlsym(g_rtc.lib, name))
  PDEOPT_RTC_SYM(create, "hiprtcCreateProgram");
  PDEOPT_RTC_SYM(compile, "hiprtcCompileProgram");
  PDEOPT_RTC_SYM(log_size, "hiprtcGetProgramLogSize");
  PDEOPT_RTC_SYM(log, "hiprtcGetProgramLog");
  PDEOPT_RTC_SYM(code_size, "hiprtcGetCodeSize");
  PDEOPT_RTC_SYM(code, "hiprtcGetCode");
  PDEOPT_RTC_SYM(destroy, "hiprtcDestroyProgram");
#undef PDEOPT_RTC_SYM
  if (!g_rtc.create || !g_rtc.compile || !g_rtc.log_size || !g_rtc.log || !g_rtc.code_size || !g_rtc.code || !g_rtc.destroy) {
    g_rtc.lib = nullptr;
    return fail(ctx, PDEOPT_EINVAL, "libhiprtc.so lacks a required symbol");
  }
  return PDEOPT_OK;
}

// compile jit_device.hpp with the two bodies for `arch`; PDEOPT_OK and the code object, or an error code and the log
template <typename T>
int compile_bodies(const std::string& mu_body, const std::string& mob_body, const std::string& arch, std::vector<char>* code, std::string* log) {
  // the bodies go in through the preprocessor: a body is one line of statements, the argument is `c`
  std::string src = std::string("#define PDEOPT_JIT_T ") + (sizeof(T) == 4 ? "float" : "double") + "\n" +
                    "#define PDEOPT_JIT_MU_BODY " + mu_body + "\n" + "#define PDEOPT_JIT_MOB_BODY " + mob_body + "\n" +
                    "#define PDEOPT_JIT_STAGEARGS_SIZE " + std::to_string(sizeof(StageArgs<T>)) + "\n" +
                    "#define PDEOPT_JIT_OFF_G " + std::to_string(offsetof(StageArgs<T>, g)) + "\n" +
                    "#define PDEOPT_JIT_OFF_EP " + std::to_string(offsetof(StageArgs<T>, ep)) + "\n" +
                    "#define PDEOPT_JIT_OFF_LC " + std::to_string(offsetof(StageArgs<T>, lc)) + "\n" +
                    "#define PDEOPT_JIT_OFF_OUT_MODE " + std::to_string(offsetof(StageArgs<T>, out_mode)) + "\n" +
                    "#define PDEOPT_JIT_ENVPARAMS_SIZE " + std::to_string(sizeof(EnvParams<T>)) + "\n" + kJitSource;
  hiprtcProgram prog = nullptr;
  if (g_rtc.create(&prog, src.c_str(), "pdeopt_jit_closures.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    *log = "hiprtcCreateProgram failed";
    return PDEOPT_EHIP;
  }
  const std::string archopt = "--offload-arch=" + arch;
  const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=fast"};
  const hiprtcResult cr = g_rtc.compile(prog, 4, opts);
  size_t n = 0;
  g_rtc.log_size(prog, &n);
  log->assign(n, '\0');
  if (n) g_rtc.log(prog, &(*log)[0]);
  if (cr != HIPRTC_SUCCESS) {
    g_rtc.destroy(&prog);
    return PDEOPT_EINVAL;
  }
  size_t cs = 0;
  g_rtc.code_size(prog, &cs);
  code->resize(cs);
  g_rtc.code(prog, code->data());
  g_rtc.destroy(&prog);
  return PDEOPT_OK;
}

template <typename T>
int get_module(pdeopt_ctx* ctx, JitModule* out) {
  const std::string key = std::to_string(ctx->device) + (sizeof(T) == 4 ? "|f32|" : "|f64|") + ctx->jit_src[0] + "|" + ctx->jit_src[1];
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_modules.find(key);
  if (it != g_modules.end()) {
    *out = it->second;
    return PDEOPT_OK;
  }
  int rc = load_rtc(ctx);
  if (rc) return rc;
  hipDeviceProp_t prop;
  PDEOPT_HIP_CHECK(ctx, hipGetDeviceProperties(&prop, ctx->device));
  std::vector<char> code;
  std::string log;
  if ((rc = compile_bodies<T>(ctx->jit_src[0], ctx->jit_src[1], prop.gcnArchName, &code, &log))) {  // "gfx950:sramecc+:xnack-": the full target id
    if (log.size() > 1500) log.resize(1500);
    return fail(ctx, rc, "run-time compilation of the closures failed (mu: `%s`; mobility: `%s`):\n%s", ctx->jit_src[0].c_str(),
                ctx->jit_src[1].c_str(), log.c_str());
  }
  JitModule m;
  PDEOPT_HIP_CHECK(ctx, hipModuleLoadData(&m.mod, code.data()));
  PDEOPT_HIP_CHECK(ctx, hipModuleGetFunction(&m.ch, m.mod, "pdeopt_jit_stage_ch"));
  PDEOPT_HIP_CHECK(ctx, hipModuleGetFunction(&m.ac, m.mod, "pdeopt_jit_stage_ac"));
  g_modules[key] = m;
  *out = m;
  return PDEOPT_OK;
}

}  // namespace

bool jit_closures_active(const pdeopt_ctx* ctx) {
  return ctx->prob.mu.kind == PDEOPT_CL_JIT || ctx->prob.mob.kind == PDEOPT_CL_JIT;
}

// one fused stencil + update launch of the run-time-compiled kernel (what launch_generic does for the closure family)
template <typename T>
int launch_jit_stage(pdeopt_ctx* ctx, const StageArgs<T>& s) {
  const pdeopt_problem& p = ctx->prob;
  if (p.equation != PDEOPT_EQ_CAHN_HILLIARD && p.equation != PDEOPT_EQ_ALLEN_CAHN)
    return fail(ctx, PDEOPT_EINVAL, "run-time-compiled closures run the 2-D Cahn-Hilliard / Allen-Cahn finite-difference kernels only");
  JitModule m;
  int rc = get_module<T>(ctx, &m);
  if (rc) return rc;
  dim3 block(64, 4, 1);
  dim3 grid((p.ny + 63) / 64, (p.nx + 3) / 4, ctx->win_n);
  if (grid.y > 65535u || grid.z > 65535u) return fail(ctx, PDEOPT_EINVAL, "grid too large for the generic kernel (nx=%d batch=%d)", p.nx, p.batch);
  StageArgs<T> args = s;
  size_t size = sizeof(args);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  PDEOPT_HIP_CHECK(ctx, hipModuleLaunchKernel(p.equation == PDEOPT_EQ_CAHN_HILLIARD ? m.ch : m.ac, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0,
                                              ctx->stream, nullptr, config));
  ctx->last_kernel = p.equation == PDEOPT_EQ_CAHN_HILLIARD ? "stage_jit<CH,hiprtc closures>" : "stage_jit<AC,hiprtc closures>";
  return PDEOPT_OK;
}

template int launch_jit_stage<float>(pdeopt_ctx*, const StageArgs<float>&);
template int launch_jit_stage<double>(pdeopt_ctx*, const StageArgs<double>&);

// pdeopt_jit_check: do these bodies compile (for gfx950, no device needed)?
int jit_check(int dtype, const char* mu_body, const char* mob_body, char* log_out, int log_cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::string log;
  int rc = PDEOPT_OK;
  if (!g_rtc.lib) {
    pdeopt_ctx tmp;
    if ((rc = load_rtc(&tmp))) log = tmp.err;
  }
  if (!rc) {
    std::vector<char> code;
    rc = dtype == PDEOPT_F32 ? compile_bodies<float>(mu_body, mob_body, "gfx950", &code, &log)
                             : compile_bodies<double>(mu_body, mob_body, "gfx950", &code, &log);
  }
  if (log_out && log_cap > 0) {
    strncpy(log_out, log.c_str(), (size_t)log_cap - 1);
    log_out[log_cap - 1] = '\0';
  }
  return rc;
}

}  // namespace pdeopt
