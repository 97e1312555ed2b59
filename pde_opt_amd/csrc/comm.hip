// RCCL communicator of the domain-decomposed driver and the substep loop that uses it (BASELINE config 5:
// "RCCL allgather of halos over xGMI only for the domain-decomposed case").
//
// One process per GPU; the host language only carries the 128-byte ncclUniqueId from rank 0 to the others
// (any transport: torch.distributed, MPI, a file).  Everything per substep runs here, with no host round trip:
//
//     for phase in plan:                         2 phases with fused stage pairs
//       pack(field)                                                    stream S
//       ncclAllGather(send, recv)                                      stream C   (waits for the pack)
//       phase(interior tiles)      -- overlaps the collective --       stream S
//       unpack(field); phase(edge tiles)                               stream S   (waits for the collective)
//
// RCCL is resolved at run time from the library the process already has (dlopen "librccl.so": torch ships
// one, /opt/rocm/lib has one), so libpdeopt_hip.so carries no link-time dependency on it and a single-GPU
// user never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <mutex>

#include "common.hpp"

// An IN-PROCESS group of ranks (pdeopt_local_group_create): the ranks are ctxs of ONE process -- several engines on
// one GPU (virtual ranks: the whole decomposed loop, neighbour tables and rank offsets into the gathered buffer run
// without a second GPU) or one engine per GPU driven by one host thread each -- and the all-gather is device-side
// copies between their strip buffers, ordered by HIP events:
//
//   rank r, exchange number q (parity p = q & 1):
//     [pack / fused pack wrote send[p]]          stream r     (before it: wait done[p][*] of exchange q - 2)
//     record ready[p][r]                         stream r
//     -- host barrier over the ranks' threads: every ready[p][*] has been recorded --
//     for every rank s:  wait ready[p][s];  copy send[p] of rank s -> recv[s] of rank r       stream r
//     record done[p][r]                          stream r
//
// Two send buffers per rank (parity) so that a rank may pack exchange q + 1 while a slower neighbour still copies
// exchange q; the wait on done[p][*] of exchange q - 2 happens-after its host-side record because this rank has
// passed the barrier of exchange q - 1, which every rank enters after recording it.  Every rank's host thread must
// be inside pdeopt_rk4_decomposed_advance at the same time (the barrier times out with an error otherwise).
struct pdeopt_local_group {
  int world = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool broken = false;
  std::vector<int> attached;          // rank taken?
  std::vector<int> device;            // [rank] HIP device of the rank's ctx
  std::vector<void*> send[2];         // [parity][rank] strip buffers, published by their owners
  std::vector<size_t> send_bytes;     // [rank]
  std::vector<hipEvent_t> ready[2], done[2];
  std::vector<int> done_recorded[2];  // has done[p][r] ever been recorded?

  // false: timed out or another rank failed
  bool barrier(double timeout_s) {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || broken; });
    if (!ok || broken) {
      broken = true;  // the other ranks must not wait for this one again
      cv.notify_all();
      return false;
    }
    return true;
  }
  void fail_all() {
    std::lock_guard<std::mutex> lk(m);
    broken = true;
    cv.notify_all();
  }
};

namespace pdeopt {

struct CommState {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0;
  hipStream_t cstream = nullptr;          // collectives
  hipEvent_t packed = nullptr, gathered = nullptr;
  void* send = nullptr;                   // strip of this rank
  void* recv = nullptr;                   // strips of all ranks, rank-major
  size_t strip_bytes = 0;
  // in-process backend (pdeopt_comm_init_local): no RCCL, copies between the ranks' buffers
  pdeopt_local_group* group = nullptr;
  void* send2 = nullptr;                  // second send buffer (parity 1)
  uint64_t seq = 0;                       // exchanges done so far
  // peer-mapped backend (pdeopt_comm_ipc_export / _attach): every rank's strip buffers are mapped into every process
  // (hipIpc), a rank's fused-pack epilogue writes its own strip, its neighbours' fused unpack reads it in place; two
  // monotone counters per rank order the exchanges, no collective
  bool ipc = false;
  void* ipc_block = nullptr;              // this rank: [strip parity 0][strip parity 1][flags: ready, consumed, error]
  size_t ipc_strip_bytes = 0;
  std::vector<void*> peer_block;          // every rank's block as mapped here (own: ipc_block)
  unsigned* ipc_err_dev = nullptr;        // time-out flag of the wait kernel (in the own block)
  bool local() const { return group != nullptr; }
  void* send_buf() const { return local() && (seq & 1) ? send2 : send; }  // where the NEXT exchange's strip goes
};

namespace {

int load_rccl(pdeopt_ctx* ctx, CommState& c) {
  if (c.lib) return PDEOPT_OK;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  // the copy the process already uses first (torch bundles its own), then the system one
  for (const char* n : names)
    if (!c.lib) c.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char* n : names)
    if (!c.lib) c.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!c.lib) return fail(ctx, PDEOPT_EINVAL, "cannot load RCCL (librccl.so): %s", dlerror());
  c.get_unique_id = reinterpret_cast<decltype(c.get_unique_id)>(dlsym(c.lib, "ncclGetUniqueId"));
  c.comm_init_rank = reinterpret_cast<decltype(c.comm_init_rank)>(dlsym(c.lib, "ncclCommInitRank"));
  c.all_gather = reinterpret_cast<decltype(c.all_gather)>(dlsym(c.lib, "ncclAllGather"));
  c.comm_destroy = reinterpret_cast<decltype(c.comm_destroy)>(dlsym(c.lib, "ncclCommDestroy"));
  c.error_string = reinterpret_cast<decltype(c.error_string)>(dlsym(c.lib, "ncclGetErrorString"));
  if (!c.get_unique_id || !c.comm_init_rank || !c.all_gather || !c.comm_destroy || !c.error_string)
    return fail(ctx, PDEOPT_EINVAL, "librccl.so lacks a required symbol");
  return PDEOPT_OK;
}

#define PDEOPT_NCCL_CHECK(ctx, c, expr)                                                                    \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess)                                                                                  \
      return fail((ctx), PDEOPT_EHIP, "%s failed: %s (%s:%d)", #expr, (c).error_string(r_), __FILE__, __LINE__); \
  } while (0)

CommState g_probe;  // symbols for pdeopt_comm_unique_id (no ctx state needed)

}  // namespace

int comm_unique_id(pdeopt_ctx* ctx, char* out128) {
  int rc = load_rccl(ctx, g_probe);
  if (rc) return rc;
  ncclUniqueId id;
  PDEOPT_NCCL_CHECK(ctx, g_probe, g_probe.get_unique_id(&id));
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return PDEOPT_OK;
}

int comm_init(pdeopt_ctx* ctx, int world, int rank, const char* id128) {
  if (!ctx->comm) ctx->comm = new CommState();
  CommState& c = *ctx->comm;
  int rc = load_rccl(ctx, c);
  if (rc) return rc;
  if (c.comm) return fail(ctx, PDEOPT_ESTATE, "this ctx already has a communicator");
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  PDEOPT_NCCL_CHECK(ctx, c, c.comm_init_rank(&c.comm, world, id, rank));
  c.world = world;
  c.rank = rank;
  PDEOPT_HIP_CHECK(ctx, hipStreamCreateWithFlags(&c.cstream, hipStreamNonBlocking));
  PDEOPT_HIP_CHECK(ctx, hipEventCreateWithFlags(&c.packed, hipEventDisableTiming));
  PDEOPT_HIP_CHECK(ctx, hipEventCreateWithFlags(&c.gathered, hipEventDisableTiming));
  return PDEOPT_OK;
}

int comm_init_local(pdeopt_ctx* ctx, pdeopt_local_group* g, int rank) {
  if (ctx->comm && (ctx->comm->comm || ctx->comm->group)) return fail(ctx, PDEOPT_ESTATE, "this ctx already has a communicator");
  {
    std::lock_guard<std::mutex> lk(g->m);
    if (rank < 0 || rank >= g->world) return fail(ctx, PDEOPT_EINVAL, "rank %d outside the local group of %d", rank, g->world);
    if (g->attached[rank]) return fail(ctx, PDEOPT_EINVAL, "rank %d of the local group is taken", rank);
    g->attached[rank] = 1;
    g->device[rank] = ctx->device;
  }
  if (!ctx->comm) ctx->comm = new CommState();
  CommState& c = *ctx->comm;
  c.group = g;
  c.world = g->world;
  c.rank = rank;
  c.seq = 0;
  for (int p = 0; p < 2; ++p) {  // the events stay with the group: a rank that re-attaches finds its own again
    if (!g->ready[p][rank]) PDEOPT_HIP_CHECK(ctx, hipEventCreateWithFlags(&g->ready[p][rank], hipEventDisableTiming));
    if (!g->done[p][rank]) PDEOPT_HIP_CHECK(ctx, hipEventCreateWithFlags(&g->done[p][rank], hipEventDisableTiming));
  }
  return PDEOPT_OK;
}

namespace {
// host-side wait for the peers' copies out of this rank's send buffers (done[p][r] of every exchange recorded so far)
void wait_peer_copies(CommState& c) {
  pdeopt_local_group& g = *c.group;
  for (int p = 0; p < 2; ++p)
    for (int r = 0; r < c.world; ++r) {
      hipEvent_t e = nullptr;
      {
        std::lock_guard<std::mutex> lk(g.m);
        if (r != c.rank && g.done_recorded[p][r]) e = g.done[p][r];
      }
      if (e) (void)hipEventSynchronize(e);
    }
}
}  // namespace

void comm_destroy(pdeopt_ctx* ctx) {
  CommState* c = ctx->comm;
  if (!c) return;
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (c->ipc) {
    for (size_t r = 0; r < c->peer_block.size(); ++r)
      if (c->peer_block[r] && c->peer_block[r] != c->ipc_block) (void)hipIpcCloseMemHandle(c->peer_block[r]);
    if (c->ipc_block) (void)hipFree(c->ipc_block);
  }
  if (c->cstream) (void)hipStreamSynchronize(c->cstream);
  if (c->comm) (void)c->comm_destroy(c->comm);
  if (c->group) {
    // peers copy OUT of this rank's send buffers on their own streams (other devices, with one engine per GPU): their
    // copies of every exchange so far must have run before the buffers are freed
    wait_peer_copies(*c);
    // the group outlives its members' buffers: un-publish them (the events stay with the group until it is destroyed)
    std::lock_guard<std::mutex> lk(c->group->m);
    c->group->send[0][c->rank] = c->group->send[1][c->rank] = nullptr;
    c->group->attached[c->rank] = 0;
    c->group->done_recorded[0][c->rank] = c->group->done_recorded[1][c->rank] = 0;
    // a group whose barrier timed out (or whose rank failed) is `broken` for good while any rank is attached: every
    // rank's call fails from then on.  It recovers once ALL ranks have detached (pdeopt_comm_destroy / ctx_destroy)
    bool any = false;
    for (int a : c->group->attached) any = any || a;
    if (!any) {
      c->group->broken = false;
      c->group->arrived = 0;
    }
  }
  if (c->send2) (void)hipFree(c->send2);
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (c->packed) (void)hipEventDestroy(c->packed);
  if (c->gathered) (void)hipEventDestroy(c->gathered);
  if (c->cstream) (void)hipStreamDestroy(c->cstream);
  delete c;
  ctx->comm = nullptr;
}

namespace {

// in-process all-gather of the strip in c.send_buf() into c.recv (header comment of this file)
int local_all_gather(pdeopt_ctx* ctx, CommState& c) {
  pdeopt_local_group& g = *c.group;
  const int p = (int)(c.seq & 1);
  PDEOPT_HIP_CHECK(ctx, hipEventRecord(g.ready[p][c.rank], ctx->stream));
  if (!g.barrier(60.0))
    return fail(ctx, PDEOPT_ESTATE, "local group: a rank did not reach exchange %llu (every rank's host thread must be "
                "inside pdeopt_rk4_decomposed_advance at the same time, with the same substep count)", (unsigned long long)c.seq);
  for (int r = 0; r < c.world; ++r) {
    void* src;
    size_t bytes;
    {
      std::lock_guard<std::mutex> lk(g.m);
      src = g.send[p][r];
      bytes = g.send_bytes[r];
    }
    if (!src || bytes != c.strip_bytes) {
      g.fail_all();
      return fail(ctx, PDEOPT_EINVAL, "local group: rank %d publishes a strip of %zu bytes, this rank expects %zu", r, bytes, c.strip_bytes);
    }
    if (r != c.rank) PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, g.ready[p][r], 0));
    PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync((char*)c.recv + (size_t)r * bytes, src, bytes, hipMemcpyDefault, ctx->stream));
  }
  PDEOPT_HIP_CHECK(ctx, hipEventRecord(g.done[p][c.rank], ctx->stream));
  {
    std::lock_guard<std::mutex> lk(g.m);
    g.done_recorded[p][c.rank] = 1;
  }
  ++c.seq;
  return PDEOPT_OK;
}

// before this rank overwrites c.send_buf() (parity p of exchange seq): every rank's copies of exchange seq - 2 out
// of that buffer must have run.  Their done events were recorded before those ranks entered the barrier of exchange
// seq - 1, which this rank has left.
int local_wait_send_free(pdeopt_ctx* ctx, CommState& c) {
  pdeopt_local_group& g = *c.group;
  const int p = (int)(c.seq & 1);
  if (c.seq < 2) return PDEOPT_OK;
  for (int r = 0; r < c.world; ++r) {
    if (r == c.rank) continue;  // own copies run on this stream, in order
    bool rec;
    {
      std::lock_guard<std::mutex> lk(g.m);
      rec = g.done_recorded[p][r] != 0;
    }
    if (rec) PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, g.done[p][r], 0));
  }
  return PDEOPT_OK;
}

}  // namespace

pdeopt_local_group* local_group_new(int world) {
  auto* g = new pdeopt_local_group();
  g->world = world;
  g->attached.assign((size_t)world, 0);
  g->device.assign((size_t)world, -1);
  g->send_bytes.assign((size_t)world, 0);
  for (int p = 0; p < 2; ++p) {
    g->send[p].assign((size_t)world, nullptr);
    g->ready[p].assign((size_t)world, nullptr);
    g->done[p].assign((size_t)world, nullptr);
    g->done_recorded[p].assign((size_t)world, 0);
  }
  return g;
}

void local_group_delete(pdeopt_local_group* g) {
  for (int p = 0; p < 2; ++p) {
    for (hipEvent_t e : g->ready[p])
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->done[p])
      if (e) (void)hipEventDestroy(e);
  }
  delete g;
}

// ------------------------------------------------------------------------------------------------------------------
// Peer-mapped exchange ("P2P stores into peer-mapped halo buffers").  One process per
// GPU as under RCCL, but no collective in the substep: every rank allocates [strip 0][strip 1][flags] once, the host
// language carries the 64-byte hipIpc handles between the processes (any transport: torch.distributed, a pipe), and
// every rank maps every other rank's block.  Exchange e lives in strip parity e & 1:
//     the kernel K_e of a rank reads strip e of its 8 neighbours IN PLACE (fused unpack through the mapped pointers) and
//     writes its own strip e + 1 (fused pack); a one-lane kernel behind it publishes  consumed = e + 1, ready = e + 2;
//     before K_e a wait kernel polls the neighbours' ready >= e + 1.
// A neighbour's ready >= e + 1 also says it has finished reading this rank's strip e - 1 (the buffer strip e + 1 goes
// into): it publishes both counters when the kernel that did both completes.  At the start of a call the first strip
// comes from the pack kernel; before it the neighbours' consumed >= e0 - 1.  The wait kernel gives up after 2 s
// (a neighbour that never arrives) and the call fails instead of hanging.
namespace {
constexpr size_t kIpcFlagBytes = 256;
struct IpcWaitArgs {
  const unsigned* flags[8];  // neighbours' flag words: [0] ready, [1] consumed
  unsigned ready_min, consumed_min;
  unsigned* err;
};
__global__ void ipc_wait_kernel(const IpcWaitArgs a) {
  const int q = threadIdx.x;
  if (q >= 8) return;
  const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
  const unsigned* f = a.flags[q];
  while (__hip_atomic_load(&f[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.ready_min ||
         __hip_atomic_load(&f[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.consumed_min) {
    if (__builtin_amdgcn_s_memrealtime() - t_in > 200000000ull) {  // 2 s of the 100 MHz clock
      __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the strips behind the counters
}
__global__ void ipc_publish_kernel(unsigned* flags, unsigned ready, unsigned consumed) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // (the stream already ordered the stencil kernel before this one)
  __hip_atomic_store(&flags[1], consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&flags[0], ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

int comm_ipc_export(pdeopt_ctx* ctx, int world, int rank, void* handle64) {
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, PDEOPT_EINVAL, "rank %d outside a world of %d", rank, world);
  if (ctx->halo != 8) return fail(ctx, PDEOPT_EINVAL, "the peer-mapped exchange runs the halo-8 layout (one exchange per substep, fused pack / unpack)");
  if (!ctx->comm) ctx->comm = new CommState();
  CommState& c = *ctx->comm;
  if (c.comm || c.group || c.ipc) return fail(ctx, PDEOPT_ESTATE, "this ctx already has a communicator");
  c.ipc_strip_bytes = halo_strip_elems(ctx) * ctx->esize;
  const size_t strip_pad = (c.ipc_strip_bytes + 255) / 256 * 256;
  PDEOPT_HIP_CHECK(ctx, hipMalloc(&c.ipc_block, 2 * strip_pad + kIpcFlagBytes));
  PDEOPT_HIP_CHECK(ctx, hipMemsetAsync(c.ipc_block, 0, 2 * strip_pad + kIpcFlagBytes, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  hipIpcMemHandle_t h;
  PDEOPT_HIP_CHECK(ctx, hipIpcGetMemHandle(&h, c.ipc_block));
  static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
  memcpy(handle64, &h, sizeof(h));
  c.world = world;
  c.rank = rank;
  c.seq = 0;
  return PDEOPT_OK;
}

int comm_ipc_attach(pdeopt_ctx* ctx, const void* handles) {
  CommState* cp = ctx->comm;
  if (!cp || !cp->ipc_block || cp->ipc) return fail(ctx, PDEOPT_ESTATE, "pdeopt_comm_ipc_export first (once per ctx)");
  CommState& c = *cp;
  c.peer_block.assign((size_t)c.world, nullptr);
  for (int r = 0; r < c.world; ++r) {
    if (r == c.rank) {
      c.peer_block[r] = c.ipc_block;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const char*>(handles) + (size_t)r * sizeof(h), sizeof(h));
    PDEOPT_HIP_CHECK(ctx, hipIpcOpenMemHandle(&c.peer_block[r], h, hipIpcMemLazyEnablePeerAccess));
  }
  const size_t strip_pad = (c.ipc_strip_bytes + 255) / 256 * 256;
  c.ipc_err_dev = reinterpret_cast<unsigned*>(static_cast<char*>(c.ipc_block) + 2 * strip_pad) + 2;
  c.ipc = true;
  return PDEOPT_OK;
}

namespace {
int rk4_ipc_advance(pdeopt_ctx* ctx, CommState& c, double dt, int64_t n, const int* nbr) {
  if (ctx->halo != 8) return fail(ctx, PDEOPT_EINVAL, "the peer-mapped exchange runs the halo-8 layout");
  if (halo_strip_elems(ctx) * ctx->esize != c.ipc_strip_bytes)
    return fail(ctx, PDEOPT_ESTATE, "the problem was reconfigured with another strip size after pdeopt_comm_ipc_export");
  int fields[4], nph = 0;
  rk4_phase_plan(ctx, fields, &nph);
  if (nph > 2) return fail(ctx, PDEOPT_EINVAL, "the peer-mapped exchange needs the fused Cahn-Hilliard kernels (one exchange per substep)");
  if (n <= 0) return PDEOPT_OK;
  const size_t strip_pad = (c.ipc_strip_bytes + 255) / 256 * 256;
  auto strip_of = [&](int rank, unsigned e) { return static_cast<char*>(c.peer_block[(size_t)rank]) + (size_t)(e & 1u) * strip_pad; };
  auto flags_of = [&](int rank) { return reinterpret_cast<unsigned*>(static_cast<char*>(c.peer_block[(size_t)rank]) + 2 * strip_pad); };
  IpcWaitArgs w{};
  for (int q = 0; q < 8; ++q) w.flags[q] = flags_of(nbr[q]);
  w.err = c.ipc_err_dev;
  unsigned e = (unsigned)c.seq;  // index of the exchange the next kernel consumes
  int rc;
  // the first strip of this call: packed from the state as it stands, into the buffer strip e - 2 lived in
  w.ready_min = 0;
  w.consumed_min = e >= 1 ? e - 1 : 0;
  hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, w);
  if ((rc = halo_pack(ctx, 0, strip_of(c.rank, e)))) return rc;
  hipLaunchKernelGGL(ipc_publish_kernel, dim3(1), dim3(1), 0, ctx->stream, flags_of(c.rank), e + 1, e);
  for (int64_t s = 0; s < n; ++s, ++e) {
    const bool more = s + 1 < n;
    w.ready_min = e + 1;
    w.consumed_min = 0;
    hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, w);
    const void* peer[8];
    for (int q = 0; q < 8; ++q) peer[q] = strip_of(nbr[q], e);
    if ((rc = rk4_substep_h8_peer(ctx, dt, more ? strip_of(c.rank, e + 1) : nullptr, peer))) return rc;
    hipLaunchKernelGGL(ipc_publish_kernel, dim3(1), dim3(1), 0, ctx->stream, flags_of(c.rank), more ? e + 2 : e + 1, e + 1);
  }
  PDEOPT_HIP_CHECK(ctx, hipGetLastError());
  c.seq = e;
  // a neighbour that never arrived: the wait kernel gave up and raised the flag (the state is then garbage)
  unsigned err = 0;
  PDEOPT_HIP_CHECK(ctx, hipMemcpyAsync(&err, c.ipc_err_dev, sizeof(err), hipMemcpyDeviceToHost, ctx->stream));
  PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (err) return fail(ctx, PDEOPT_ESTATE, "peer-mapped exchange: a neighbour rank did not publish its strip within 2 s");
  return PDEOPT_OK;
}
}  // namespace

// n RK4 substeps of this rank's tile; nbr[8] = ranks of {up, down, left, right, UL, UR, DL, DR}
int rk4_decomposed_advance(pdeopt_ctx* ctx, double dt, int64_t n, const int* nbr, int overlap) {
  CommState* cp = ctx->comm;
  if (!cp || (!cp->comm && !cp->group && !cp->ipc))
    return fail(ctx, PDEOPT_ESTATE, "pdeopt_comm_init / pdeopt_comm_init_local / pdeopt_comm_ipc_attach has not been called");
  CommState& c = *cp;
  if (c.ipc) {
    for (int q = 0; q < 8; ++q)
      if (nbr[q] < 0 || nbr[q] >= c.world) return fail(ctx, PDEOPT_EINVAL, "neighbour rank %d outside 0..%d", nbr[q], c.world - 1);
    return rk4_ipc_advance(ctx, c, dt, n, nbr);
  }
  for (int q = 0; q < 8; ++q)
    if (nbr[q] < 0 || nbr[q] >= c.world) return fail(ctx, PDEOPT_EINVAL, "neighbour rank %d outside 0..%d", nbr[q], c.world - 1);
  const size_t strip_elems = halo_strip_elems(ctx);
  const size_t strip_bytes = strip_elems * ctx->esize;
  int rc;
  if (c.strip_bytes != strip_bytes) {
    PDEOPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (c.local()) {  // nobody may still be copying out of the old buffers: wait for the peers' last copies
      wait_peer_copies(c);
      std::lock_guard<std::mutex> lk(c.group->m);
      c.group->send[0][c.rank] = c.group->send[1][c.rank] = nullptr;
    }
    if (c.send) (void)hipFree(c.send);
    if (c.send2) (void)hipFree(c.send2);
    if (c.recv) (void)hipFree(c.recv);
    c.send = c.send2 = c.recv = nullptr;
    if ((rc = ensure_buffer(ctx, &c.send, strip_bytes))) return rc;
    if (c.local() && (rc = ensure_buffer(ctx, &c.send2, strip_bytes))) return rc;
    if ((rc = ensure_buffer(ctx, &c.recv, strip_bytes * (size_t)c.world))) return rc;
    c.strip_bytes = strip_bytes;
    if (c.local()) {
      std::lock_guard<std::mutex> lk(c.group->m);
      c.group->send[0][c.rank] = c.send;
      c.group->send[1][c.rank] = c.send2;
      c.group->send_bytes[c.rank] = strip_bytes;
    }
  }
  int fields[4], nph = 0;
  rk4_phase_plan(ctx, fields, &nph);
  const ncclDataType_t dtype = ctx->prob.dtype == PDEOPT_F32 ? ncclFloat32 : ncclFloat64;
  // the collective: all ranks' strips (this rank's in `send`) -> c.recv, rank-major, on `stream`
  auto all_gather = [&](void* send, hipStream_t stream) -> int {
    if (c.local()) return local_all_gather(ctx, c);  // (always on the compute stream)
    PDEOPT_NCCL_CHECK(ctx, c, c.all_gather(send, c.recv, strip_elems, dtype, c.comm, stream));
    return PDEOPT_OK;
  };
  auto abort_group = [&](int code) {
    if (c.local()) c.group->fail_all();  // the other ranks' threads must not wait for this one
    return code;
  };

  // In-process ranks on ONE device: no copies at all.  A rank's kernel reads its neighbours' strips where their kernels
  // wrote them (the fused unpack through the neighbours' OWN buffers, as in the peer-mapped exchange above) -- per substep
  // and rank one kernel, the waits on the neighbours' `ready` events and one event record, instead of world copies of
  // a strip into a gathered buffer.  Exchange e lives in send[e & 1]; the kernel that reads exchange e writes e + 1:
  //   * it may read a neighbour's strip e once that neighbour's `ready` of e is reached (recorded behind the kernel /
  //     the pack that wrote it; the host barrier orders the record before this rank's wait is enqueued);
  //   * it overwrites this rank's strip e - 1, which the neighbours' kernels e - 1 read: those precede the neighbours'
  //     `ready` of e in their streams -- the same wait.
  // `done` (recorded behind every kernel) is what comm_destroy waits for before the buffers go.
  bool same_device = c.local();
  if (c.local()) {
    std::lock_guard<std::mutex> lk(c.group->m);
    for (int r = 0; r < c.world; ++r) same_device = same_device && c.group->device[r] == ctx->device;
  }
  if (ctx->halo == 8 && same_device && getenv("PDEOPT_LOCAL_GATHER") == nullptr) {
    if (n <= 0) return PDEOPT_OK;
    pdeopt_local_group& g = *c.group;
    auto publish = [&]() -> int {  // exchange c.seq is complete in this rank's stream
      const int p = (int)(c.seq & 1);
      PDEOPT_HIP_CHECK(ctx, hipEventRecord(g.ready[p][c.rank], ctx->stream));
      if (!g.barrier(60.0))
        return fail(ctx, PDEOPT_ESTATE, "local group: a rank did not reach exchange %llu (every rank's host thread must be "
                    "inside pdeopt_rk4_decomposed_advance at the same time, with the same substep count)", (unsigned long long)c.seq);
      ++c.seq;
      return PDEOPT_OK;
    };
    if ((rc = halo_pack(ctx, 0, c.send_buf()))) return abort_group(rc);
    if ((rc = publish())) return abort_group(rc);
    for (int64_t s = 0; s < n; ++s) {
      const bool more = s + 1 < n;
      const int p = (int)((c.seq - 1) & 1);  // parity of the exchange this substep reads
      const void* peer[8];
      {
        std::lock_guard<std::mutex> lk(g.m);
        for (int q = 0; q < 8; ++q) {
          peer[q] = g.send[p][nbr[q]];
          if (!peer[q] || g.send_bytes[nbr[q]] != c.strip_bytes) {
            g.broken = true;
            g.cv.notify_all();
            return fail(ctx, PDEOPT_EINVAL, "local group: rank %d publishes a strip of %zu bytes, this rank expects %zu", nbr[q],
                        g.send_bytes[nbr[q]], c.strip_bytes);
          }
        }
      }
      for (int q = 0; q < 8; ++q) {
        bool seen = nbr[q] == c.rank;
        for (int q2 = 0; q2 < q; ++q2) seen = seen || nbr[q2] == nbr[q];
        if (!seen) PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, g.ready[p][nbr[q]], 0));
      }
      if ((rc = rk4_substep_h8_peer(ctx, dt, more ? c.send_buf() : nullptr, peer))) return abort_group(rc);
      PDEOPT_HIP_CHECK(ctx, hipEventRecord(g.done[p][c.rank], ctx->stream));
      {
        std::lock_guard<std::mutex> lk(g.m);
        g.done_recorded[p][c.rank] = 1;
      }
      if (more && (rc = publish())) return abort_group(rc);
    }
    return PDEOPT_OK;
  }

  if (ctx->halo == 8) {
    // ONE exchange per substep (halo-8 layout):
    //   pack(Y) -> all-gather                                   prologue: the state as it stands
    //   per substep:  PAIR_12 on tile + 4, its edge tiles reading the halo from the gathered strips (no unpack
    //                 launch) -> PAIR_34, its edge tiles writing the NEW state's strip (no pack launch)
    //                 -> all-gather (not after the last substep)
    if (n <= 0) return PDEOPT_OK;
    if (c.local() && (rc = local_wait_send_free(ctx, c))) return abort_group(rc);
    void* send = c.send_buf();
    if ((rc = halo_pack(ctx, 0, send))) return abort_group(rc);
    if ((rc = all_gather(send, ctx->stream))) return abort_group(rc);
    for (int64_t s = 0; s < n; ++s) {
      const bool more = s + 1 < n;
      if (more && c.local() && (rc = local_wait_send_free(ctx, c))) return abort_group(rc);
      send = c.send_buf();
      // no unpack launch: the first pair's edge tiles read the halo from c.recv (and fill the field's frame)
      if ((rc = rk4_substep_h8(ctx, dt, more ? send : nullptr, c.recv, nbr))) return abort_group(rc);
      if (more && (rc = all_gather(send, ctx->stream))) return abort_group(rc);
    }
    return PDEOPT_OK;
  }

  const bool split = overlap && nph == 2 && !c.local();  // interior / edge launches exist for the fused stage pairs
  for (int64_t s = 0; s < n; ++s) {
    for (int ph = 0; ph < nph; ++ph) {
      if (c.local() && (rc = local_wait_send_free(ctx, c))) return abort_group(rc);
      void* send = c.send_buf();
      if ((rc = halo_pack(ctx, fields[ph], send))) return abort_group(rc);
      if (split) {
        PDEOPT_HIP_CHECK(ctx, hipEventRecord(c.packed, ctx->stream));
        PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(c.cstream, c.packed, 0));
        if ((rc = all_gather(send, c.cstream))) return rc;
        PDEOPT_HIP_CHECK(ctx, hipEventRecord(c.gathered, c.cstream));
        if ((rc = rk4_phase(ctx, ph, dt, 1))) return rc;  // interior tiles: no halo reads
        PDEOPT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, c.gathered, 0));
        if ((rc = halo_unpack(ctx, fields[ph], c.recv, nbr))) return rc;
        if ((rc = rk4_phase(ctx, ph, dt, 2))) return rc;  // edge tiles
      } else {
        if ((rc = all_gather(send, ctx->stream))) return abort_group(rc);
        if ((rc = halo_unpack(ctx, fields[ph], c.recv, nbr))) return abort_group(rc);
        if ((rc = rk4_phase(ctx, ph, dt, 0))) return abort_group(rc);
      }
    }
  }
  return PDEOPT_OK;
}

}  // namespace pdeopt
