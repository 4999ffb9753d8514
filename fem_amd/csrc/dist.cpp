// dist.cpp -- libfem_dist.so: the multi-GPU stepping loop over the C-ABI of libfem_hip.so (include/fem_hip.h), the
// HIP runtime (streams, events) and RCCL (ncclSend / ncclRecv).  See include/fem_dist.h.
//
// Schedule (the same one fem_amd/dist.py:ShardedRun drives through torch.distributed; DESIGN.md "Multi-GPU"):
// the tiles of a shard's fused plan come as [interior | boundary | ghost].  Steps run in groups of at most ksteps;
// the group that uses up the ghost budget runs the boundary tiles first, packs their results from the NEXT
// buffer, starts the exchange into the ghost segment of that buffer on the communication stream, and lets the
// interior tiles run beside it (ghost tiles are skipped: the exchange overwrites them).  The first group after an
// exchange is one launch of everything when the exchange has already arrived, otherwise the interior tiles go
// first and the compute stream waits for the exchange behind them.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fem_dist.h"

namespace {

thread_local std::string g_detail;
int fail(int status, const std::string &what) {
  g_detail = what;
  return status;
}
#define HIP_OK(call)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) return fail(FEM_ERR_DEVICE, std::string(#call ": ") + hipGetErrorString(e_)); \
  } while (0)
#define NCCL_OK(call)                                                                                     \
  do {                                                                                                    \
    ncclResult_t r_ = (call);                                                                             \
    if (r_ != ncclSuccess) return fail(FEM_ERR_DEVICE, std::string(#call ": ") + ncclGetErrorString(r_)); \
  } while (0)

#define FEM_TRY(call)            \
  do {                           \
    const int rc_ = (call);      \
    if (rc_ != FEM_OK) return rc_; \
  } while (0)

// Reusable barrier for the threads of an in-process world.  A rank that leaves its loop with an error breaks the
// barrier (abort()), and a rank that waits longer than the limit (FEM_DIST_BARRIER_SECONDS, default 60) breaks it
// too: the others then return an error instead of waiting for a rank that will not come.
class Barrier {
 public:
  explicit Barrier(int n) : n_(n) {
    const char *e = getenv("FEM_DIST_BARRIER_SECONDS");
    if (e && atof(e) > 0) limit_ = atof(e);
  }
  bool wait() {                          // false: broken
    std::unique_lock<std::mutex> lock(m_);
    if (broken_) return false;
    const int gen = gen_;
    if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); return true; }
    const bool ok = cv_.wait_for(lock, std::chrono::duration<double>(limit_), [&] { return gen != gen_ || broken_; });
    if (!ok) { broken_ = true; cv_.notify_all(); }
    return !broken_;
  }
  void abort() {
    std::lock_guard<std::mutex> lock(m_);
    broken_ = true;
    cv_.notify_all();
  }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, count_ = 0, gen_ = 0;
  bool broken_ = false;
  double limit_ = 60.0;
};

const int g_failRank = getenv("FEM_DIST_TEST_FAIL_RANK") ? atoi(getenv("FEM_DIST_TEST_FAIL_RANK")) : -1;
const bool g_trace = getenv("FEM_DIST_TRACE") && strcmp(getenv("FEM_DIST_TRACE"), "0") != 0;   // one stderr line per stage
#define TRACE(d, ...)                                              \
  do {                                                             \
    if (g_trace) {                                                 \
      char b_[160];                                                \
      std::snprintf(b_, sizeof b_, __VA_ARGS__);                   \
      std::fprintf(stderr, "[fem_dist %d] %s\n", (d)->rank, b_);   \
    }                                                              \
  } while (0)

}  // namespace

struct fem_dist_world {
  int nranks;
  Barrier barrier;
  std::vector<fem_dist *> ranks;
  explicit fem_dist_world(int n) : nranks(n), barrier(n), ranks(n, nullptr) {}
};

struct fem_dist {
  fem_plan *plan = nullptr;
  int rank = 0, nranks = 1, krings = 1, ksteps = 0, sb = 4;
  fem_uint nI = 0, nB = 0, nG = 0, nOwn = 0, nLocal = 0, nSend = 0;
  bool classes = false;                  // fused plan run with the overlapped schedule
  int mode = 1;                          // FEM_OVERLAP: 0 none, 1 adaptive, 2 split, 3 sync
  std::vector<int> sc, rc;               // vertices to / from each rank
  std::vector<size_t> soff, roff;        // their offsets (vertices) in the send buffer / ghost segment
  void *sendbuf = nullptr;
  hipStream_t comm = nullptr, side = nullptr;
  hipEvent_t evPacked = nullptr, evDone = nullptr, evTop = nullptr, evSide = nullptr;
  bool sideBusy = false;
  ncclComm_t nccl = nullptr;
  fem_dist_world *world = nullptr;
  double hostVals[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // in-process all-reduce staging
  int budget = 0;
  bool pending = false, everExchanged = false;
  int nExchanges = 0, nMerged = 0, nSplit = 0;
};

namespace {

size_t pairBytes(const fem_dist *d) { return 2 * (size_t)d->sb; }

// One all-to-all-v of boundary state from state buffer `which` (0 current, 1 next) into the ghost segments of
// the neighbours' buffers of the same kind.  Pack on `stream`; everything else on the communication stream,
// ordered behind the pack by an event; evDone marks the arrival of this rank's ghosts.
// Host-side wait for an event / a stream by polling (hipEventQuery / hipStreamQuery) with a limit, so that work
// that never completes becomes an error message instead of a thread asleep in the runtime (FEM_DIST_WAIT=block:
// hipEventSynchronize / hipStreamSynchronize instead).
const bool g_blockingWait = getenv("FEM_DIST_WAIT") && !strcmp(getenv("FEM_DIST_WAIT"), "block");
double waitLimitSeconds() {
  const char *e = getenv("FEM_DIST_BARRIER_SECONDS");
  return e && atof(e) > 0 ? atof(e) : 60.0;
}
template <typename Query> int pollUntilDone(Query query, const char *what) {
  const auto t0 = std::chrono::steady_clock::now();
  const double limit = waitLimitSeconds();
  for (unsigned spins = 0;; spins++) {
    const hipError_t e = query();
    if (e == hipSuccess) return FEM_OK;
    if (e != hipErrorNotReady) return fail(FEM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else std::this_thread::yield();
    if ((spins & 255) == 255 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
      return fail(FEM_ERR_DEVICE, std::string(what) + ": not complete after " + std::to_string((int)limit) + " s");
  }
}
int hostWaitEvent(hipEvent_t ev, const char *what) {
  if (g_blockingWait) { HIP_OK(hipEventSynchronize(ev)); return FEM_OK; }
  return pollUntilDone([&] { return hipEventQuery(ev); }, what);
}

int brokenBarrier(const fem_dist *d, const char *where) {
  return fail(FEM_ERR_DEVICE, std::string("in-process world: barrier broken at ") + where + " of rank " + std::to_string(d->rank) +
                                  " (another rank failed, or did not arrive within the limit)");
}

// In-process worlds order the ranks' copies on the device (default: streams wait for the other ranks' events, the
// host only meets at barriers so that every event is recorded before it is waited for) or on the host
// (FEM_DIST_INPROC=host: a rank waits for its own events and the ranks meet at a barrier, so no stream ever waits for
// an event that another thread recorded -- a diagnostic).
const bool g_inprocEvents = !(getenv("FEM_DIST_INPROC") && !strcmp(getenv("FEM_DIST_INPROC"), "host"));

int startExchange(fem_dist *d, int which, hipStream_t stream, uint64_t vec = 0) {     // vec != 0: that vector instead of a state buffer
  TRACE(d, "exchange %d: begin", d->nExchanges);
  if (g_failRank == d->rank && d->nExchanges >= 1)        // fault injection for the tests (FEM_DIST_TEST_FAIL_RANK)
    return fail(FEM_ERR_DEVICE, "injected failure of rank " + std::to_string(d->rank) + " (FEM_DIST_TEST_FAIL_RANK)");
  if (d->world) {
    // the previous exchange's copies out of every send buffer must be done before it is packed again
    if (!g_inprocEvents && d->everExchanged) {
      TRACE(d, "exchange %d: waiting for the copies of the previous one", d->nExchanges);
      FEM_TRY(hostWaitEvent(d->evDone, "the copies of the previous exchange"));     // this rank's copies out of the others' buffers
    }
    if (!d->world->barrier.wait()) return brokenBarrier(d, "the start of an exchange");
    if (g_inprocEvents && d->everExchanged)
      for (fem_dist *p : d->world->ranks)
        if (p != d && p->rc[d->rank] > 0) HIP_OK(hipStreamWaitEvent(stream, p->evDone, 0));
  }
  int rc = vec ? fem_plan_pack_vector(d->plan, vec, (uint64_t)d->sendbuf, stream)
               : fem_plan_pack_halo_from(d->plan, which, (uint64_t)d->sendbuf, stream);
  if (rc != FEM_OK) return rc;
  HIP_OK(hipEventRecord(d->evPacked, stream));
  HIP_OK(hipStreamWaitEvent(d->comm, d->evPacked, 0));
  uint64_t buf = vec;
  if (!vec) rc = fem_plan_state_buffer(d->plan, which, &buf);
  if (rc != FEM_OK) return rc;
  char *recv = (char *)buf + (size_t)d->nOwn * pairBytes(d);
  if (d->world) {
    if (!g_inprocEvents) {
      TRACE(d, "exchange %d: waiting for the pack", d->nExchanges);
      FEM_TRY(hostWaitEvent(d->evPacked, "the pack of an exchange"));     // this rank's send buffer is filled
    }
    TRACE(d, "exchange %d: packed", d->nExchanges);
    if (!d->world->barrier.wait()) return brokenBarrier(d, "the pack of an exchange");   // every rank has packed (events: recorded its pack event)
    for (int r = 0; r < d->nranks; r++) {
      if (r == d->rank || d->rc[r] == 0) continue;
      fem_dist *p = d->world->ranks[r];
      if (g_inprocEvents) HIP_OK(hipStreamWaitEvent(d->comm, p->evPacked, 0));
      HIP_OK(hipMemcpyAsync(recv + d->roff[r] * pairBytes(d), (const char *)p->sendbuf + p->soff[d->rank] * pairBytes(d),
                            (size_t)d->rc[r] * pairBytes(d), hipMemcpyDeviceToDevice, d->comm));
    }
  } else {
    const ncclDataType_t ty = d->sb == 4 ? ncclFloat32 : ncclFloat64;
    NCCL_OK(ncclGroupStart());
    for (int r = 0; r < d->nranks; r++) {
      if (r == d->rank) continue;
      if (d->sc[r] > 0)
        NCCL_OK(ncclSend((const char *)d->sendbuf + d->soff[r] * pairBytes(d), 2 * (size_t)d->sc[r], ty, r, d->nccl, d->comm));
      if (d->rc[r] > 0) NCCL_OK(ncclRecv(recv + d->roff[r] * pairBytes(d), 2 * (size_t)d->rc[r], ty, r, d->nccl, d->comm));
    }
    NCCL_OK(ncclGroupEnd());
  }
  HIP_OK(hipEventRecord(d->evDone, d->comm));
  TRACE(d, "exchange %d: issued", d->nExchanges);
  d->pending = true;
  d->everExchanged = true;
  d->nExchanges++;
  return FEM_OK;
}

int waitExchange(fem_dist *d, hipStream_t stream) {
  if (d->pending) HIP_OK(hipStreamWaitEvent(stream, d->evDone, 0));
  d->pending = false;
  return FEM_OK;
}

int runOverlapped(fem_dist *d, int nsteps, double dt, const fem_model *model, hipStream_t stream) {
  const fem_uint nI = d->nI, nB = d->nB, nG = d->nG;
  fem_plan *p = d->plan;
  hipStream_t side = d->side;             // stream of the boundary tiles + pack (null: all on `stream`)
  for (int done = 0; done < nsteps;) {
    const int s = std::min(std::min(d->ksteps, d->budget), nsteps - done);
    const bool first = d->pending, last = d->budget == s;
    bool merged = false;
    if (side) {
      HIP_OK(hipEventRecord(d->evTop, stream));                     // everything of the groups before this one
      if (d->sideBusy) { HIP_OK(hipStreamWaitEvent(stream, d->evSide, 0)); d->sideBusy = false; }   // ... their boundary tiles too
    }
    if (first) {
      // the exchange has had the previous group's interior launch to finish: if it has, this group is ONE
      // launch (no short boundary launch with its tail); if not, the interior tiles go first
      const bool arrived = d->mode != 2 && hipEventQuery(d->evDone) == hipSuccess;
      if (arrived) { merged = true; d->nMerged++; }
      else { FEM_TRY(fem_plan_launch_tiles(p, 0, nI, s, dt, model, stream)); d->nSplit++; }
      if (side && last) HIP_OK(hipStreamWaitEvent(side, d->evDone, 0));   // the boundary tiles read the ghost rows too
      FEM_TRY(waitExchange(d, stream));
    }
    if (last) {
      // boundary tiles, pack and exchange on the side stream, the interior tiles of the same group at once on
      // `stream`: the two launches share the chip instead of a short launch with its tail in front of a long one
      hipStream_t bs = side ? side : stream;
      if (side) HIP_OK(hipStreamWaitEvent(side, d->evTop, 0));
      FEM_TRY(fem_plan_launch_tiles(p, nI, nB, s, dt, model, bs));
      FEM_TRY(startExchange(d, 1, bs));
      if (side) { HIP_OK(hipEventRecord(d->evSide, side)); d->sideBusy = true; }
      if (d->mode == 3) HIP_OK(hipEventSynchronize(d->evDone));
      if (!first || merged) FEM_TRY(fem_plan_launch_tiles(p, 0, nI, s, dt, model, stream));
      d->budget = d->krings;
    } else {
      if (first && !merged) FEM_TRY(fem_plan_launch_tiles(p, nI, nB + nG, s, dt, model, stream));
      else FEM_TRY(fem_plan_launch_tiles(p, 0, nI + nB + nG, s, dt, model, stream));
      d->budget -= s;
    }
    FEM_TRY(fem_plan_flip(p));
    done += s;
  }
  if (side && d->sideBusy) { HIP_OK(hipStreamWaitEvent(stream, d->evSide, 0)); d->sideBusy = false; }   // `stream` ends behind everything issued
  return FEM_OK;
}

int runPlain(fem_dist *d, int nsteps, double dt, const fem_model *model, hipStream_t stream) {
  for (int done = 0; done < nsteps;) {
    if (d->budget == 0) {
      FEM_TRY(startExchange(d, 0, stream));
      FEM_TRY(waitExchange(d, stream));
      d->budget = d->krings;
    }
    const int chunk = std::min(d->budget, nsteps - done);
    FEM_TRY(fem_plan_run(d->plan, chunk, dt, model, stream));
    d->budget -= chunk;
    done += chunk;
  }
  return FEM_OK;
}

// the two communication steps of the distributed conjugate gradients (fem_plan_run_imex_sharded)
int imexExchange(void *ctx, uint64_t vec, void *stream) {
  fem_dist *d = (fem_dist *)ctx;
  FEM_TRY(startExchange(d, 0, (hipStream_t)stream, vec));
  return waitExchange(d, (hipStream_t)stream);
}

int imexAllreduce(void *ctx, uint64_t dev, int count, void *stream) {
  fem_dist *d = (fem_dist *)ctx;
  hipStream_t st = (hipStream_t)stream;
  if (!d->world) {
    NCCL_OK(ncclAllReduce((const void *)dev, (void *)dev, (size_t)count, ncclDouble, ncclSum, d->nccl, st));
    return FEM_OK;
  }
  // in-process world: through the host, summed in rank order on every rank (identical results on all of them)
  if (count > 8) return fail(FEM_ERR_INVALID, "imexAllreduce: at most 8 values");
  HIP_OK(hipMemcpyAsync(d->hostVals, (const void *)dev, sizeof(double) * count, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  if (!d->world->barrier.wait()) return brokenBarrier(d, "an all-reduce");
  double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (fem_dist *p : d->world->ranks)
    for (int i = 0; i < count; i++) sum[i] += p->hostVals[i];
  if (!d->world->barrier.wait()) return brokenBarrier(d, "an all-reduce");   // nobody overwrites its values before everybody has read them
  std::memcpy(d->hostVals, sum, sizeof(double) * count);
  HIP_OK(hipMemcpyAsync((void *)dev, d->hostVals, sizeof(double) * count, hipMemcpyHostToDevice, st));
  HIP_OK(hipStreamSynchronize(st));
  return FEM_OK;
}

}  // namespace

extern "C" {

int fem_dist_run_imex(fem_dist *d, int nsteps, double dt, const fem_model *model, int cg_iterations, void *stream) {
  if (!d) return fail(FEM_ERR_INVALID, "fem_dist_run_imex: null");
  g_detail.clear();
  fem_imex_comm comm = {d, imexExchange, imexAllreduce};
  const int rc = fem_plan_run_imex_sharded(d->plan, nsteps, dt, model, cg_iterations, d->nOwn, &comm, stream);
  if (rc != FEM_OK && g_detail.empty()) g_detail = fem_error_detail();
  if (rc != FEM_OK && d->world) d->world->barrier.abort();     // the other ranks must not wait for this one
  return rc;
}

int fem_dist_run_imex_fused(fem_dist *d, int nsteps, double dt, const fem_model *model, double rel_tol, double wsum_max, void *stream) {
  if (!d) return fail(FEM_ERR_INVALID, "fem_dist_run_imex_fused: null");
  g_detail.clear();
  // a launch takes up to ksteps iterations on the ghost rows as they stand: the shard needs at least that many rings
  if (d->ksteps < 1 || d->krings < d->ksteps)
    return fail(FEM_ERR_INVALID, "fem_dist_run_imex_fused: needs a fused-tile plan whose fused depth does not exceed the shard's ghost rings");
  fem_imex_comm comm = {d, imexExchange, imexAllreduce};       // the Chebyshev solver calls the exchange only
  const int rc = fem_plan_run_imex_fused_sharded(d->plan, nsteps, dt, model, rel_tol, wsum_max, &comm, stream);
  if (rc != FEM_OK && g_detail.empty()) g_detail = fem_error_detail();
  if (rc != FEM_OK && d->world) d->world->barrier.abort();
  return rc;
}

int fem_dist_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int fem_dist_set_device(int device) {
  HIP_OK(hipSetDevice(device));
  return FEM_OK;
}

int fem_dist_stream_create(void **stream) {
  if (!stream) return fail(FEM_ERR_INVALID, "fem_dist_stream_create: null");
  hipStream_t s = nullptr;
  HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return FEM_OK;
}

void fem_dist_stream_destroy(void *stream) {
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
}

int fem_dist_stream_synchronize(void *stream) {
  if (g_blockingWait) { HIP_OK(hipStreamSynchronize((hipStream_t)stream)); return FEM_OK; }
  return pollUntilDone([&] { return hipStreamQuery((hipStream_t)stream); }, "fem_dist_stream_synchronize");
}

int fem_dist_rccl_unique_id(char id_out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_OK(ncclGetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return FEM_OK;
}

int fem_dist_rccl_comm_create(const char id[128], int rank, int nranks, void **comm) {
  if (!id || !comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(FEM_ERR_INVALID, "fem_dist_rccl_comm_create: bad argument");
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclComm_t c = nullptr;
  NCCL_OK(ncclCommInitRank(&c, nranks, uid, rank));
  *comm = c;
  return FEM_OK;
}

void fem_dist_rccl_comm_destroy(void *comm) {
  if (comm) (void)ncclCommDestroy((ncclComm_t)comm);
}

int fem_dist_world_create(int nranks, fem_dist_world **out) {
  if (nranks < 1 || !out) return fail(FEM_ERR_INVALID, "fem_dist_world_create: bad argument");
  *out = new fem_dist_world(nranks);
  return FEM_OK;
}

void fem_dist_world_destroy(fem_dist_world *w) { delete w; }

void fem_dist_destroy(fem_dist *d) {
  if (!d) return;
  if (d->world && d->world->ranks[d->rank] == d) d->world->ranks[d->rank] = nullptr;
  if (d->comm) { (void)hipStreamSynchronize(d->comm); (void)hipStreamDestroy(d->comm); }
  if (d->side) { (void)hipStreamSynchronize(d->side); (void)hipStreamDestroy(d->side); }
  if (d->evTop) (void)hipEventDestroy(d->evTop);
  if (d->evSide) (void)hipEventDestroy(d->evSide);
  if (d->evPacked) (void)hipEventDestroy(d->evPacked);
  if (d->evDone) (void)hipEventDestroy(d->evDone);
  if (d->sendbuf) (void)hipFree(d->sendbuf);
  delete d;
}

int fem_dist_create(fem_plan *plan, const fem_shard *shard, void *comm, fem_dist_world *world, fem_dist **out) {
  if (!plan || !shard || !out || (comm == nullptr) == (world == nullptr))
    return fail(FEM_ERR_INVALID, "fem_dist_create: plan, shard and exactly one of comm / world are needed");
  fem_dist *d = new fem_dist();
  d->plan = plan;
  d->nccl = (ncclComm_t)comm;
  d->world = world;
  fem_uint ownB = 0, ownE = 0;
  fem_shard_info(shard, &ownB, &ownE, &d->nLocal, &d->nSend);
  d->nOwn = ownE - ownB;
  const fem_uint *l2g = nullptr, *sidx = nullptr;
  const int *sc = nullptr, *rc = nullptr;
  fem_shard_arrays(shard, &l2g, &sidx, &sc, &rc);
  fem_shard_ranks(shard, &d->rank, &d->nranks, &d->krings);
  if (comm) {
    int n = 0, r = 0;
    if (ncclCommCount(d->nccl, &n) != ncclSuccess || ncclCommUserRank(d->nccl, &r) != ncclSuccess || n != d->nranks || r != d->rank) {
      delete d;
      return fail(FEM_ERR_INVALID, "fem_dist_create: the communicator's rank / size differ from the shard's");
    }
  } else if (world->nranks != d->nranks || world->ranks[d->rank] != nullptr) {
    delete d;
    return fail(FEM_ERR_INVALID, "fem_dist_create: the world's size differs from the shard's, or the rank is taken");
  }
  uint64_t nv = 0;
  int sbytes = 0, ks = 0;
  int rcode = fem_plan_state(plan, nullptr, &nv, &sbytes, nullptr, nullptr);
  if (rcode == FEM_OK) rcode = fem_plan_tile_info(plan, &ks, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rcode != FEM_OK || nv != d->nLocal) {
    delete d;
    return fail(FEM_ERR_INVALID, "fem_dist_create: the plan is not a plan of this shard");
  }
  d->sb = sbytes; d->ksteps = ks;
  d->budget = d->krings;                  // ghosts are fresh: k steps allowed
  d->sc.assign(sc, sc + d->nranks); d->rc.assign(rc, rc + d->nranks);
  d->soff.assign(d->nranks, 0); d->roff.assign(d->nranks, 0);
  size_t so = 0, ro = 0;
  for (int r = 0; r < d->nranks; r++) { d->soff[r] = so; d->roff[r] = ro; so += (size_t)d->sc[r]; ro += (size_t)d->rc[r]; }
  if (so != d->nSend || ro != d->nLocal - d->nOwn) {
    delete d;
    return fail(FEM_ERR_INVALID, "fem_dist_create: inconsistent shard counts");
  }
  const char *ov = getenv("FEM_OVERLAP");
  d->mode = !ov ? 1 : (!strcmp(ov, "0") ? 0 : (!strcmp(ov, "split") ? 2 : (!strcmp(ov, "sync") ? 3 : 1)));
  if (ks > 0 && d->mode != 0) {
    if (fem_plan_tile_classes(plan, &d->nI, &d->nB, &d->nG) == FEM_OK) d->classes = true;
  }
  static std::mutex setupMutex;            // ranks of an in-process world create their streams one after the other
  std::lock_guard<std::mutex> setupLock(setupMutex);
  hipError_t e = hipStreamCreateWithFlags(&d->comm, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&d->evPacked, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&d->evDone, hipEventDisableTiming);
  const char *bsEnv = getenv("FEM_BOUNDARY_STREAM");
  if (e == hipSuccess && d->classes && !(bsEnv && !strcmp(bsEnv, "0"))) {
    // Normal priority, always ("the side-stream stall"): a side stream at the greatest priority
    // gained nothing measurable (8-GPU rehearsal 6.20x against 6.21-6.24x) and, with three or four ranks of one process
    // each holding one, starved the normal-priority streams its own waits depended on -- 1 run in ~17 never finished,
    // with 16 hardware queues as with 4, with host-ordered copies as with event-ordered ones; 0 of 240 at normal
    // priority.  The knob that brought it back (FEM_SIDE_PRIORITY) is gone.
    e = hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->evTop, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->evSide, hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipMalloc(&d->sendbuf, std::max<size_t>(1, d->nSend) * pairBytes(d));
  if (e != hipSuccess) {
    fem_dist_destroy(d);
    return fail(FEM_ERR_DEVICE, std::string("fem_dist_create: ") + hipGetErrorString(e));
  }
  if (world) world->ranks[d->rank] = d;
  *out = d;
  return FEM_OK;
}

const char *fem_dist_error_detail(void) { return g_detail.c_str(); }

int fem_dist_run(fem_dist *d, int nsteps, double dt, const fem_model *model, void *stream) {
  if (!d || nsteps < 0) return fail(FEM_ERR_INVALID, "fem_dist_run: bad argument");
  g_detail.clear();
  TRACE(d, "run %d steps (budget %d)", nsteps, d->budget);
  const int rc = d->classes ? runOverlapped(d, nsteps, dt, model, (hipStream_t)stream) : runPlain(d, nsteps, dt, model, (hipStream_t)stream);
  if (rc != FEM_OK && g_detail.empty()) g_detail = fem_error_detail();
  if (rc != FEM_OK && d->world) d->world->barrier.abort();     // the other ranks must not wait for this one
  TRACE(d, "run: queued, status %d", rc);
  return rc;
}

int fem_dist_wait(fem_dist *d, void *stream) {
  if (!d) return fail(FEM_ERR_INVALID, "fem_dist_wait: null");
  return waitExchange(d, (hipStream_t)stream);
}

void fem_dist_stats(const fem_dist *d, int *exchanges, int *merged_groups, int *split_groups) {
  if (exchanges) *exchanges = d ? d->nExchanges : 0;
  if (merged_groups) *merged_groups = d ? d->nMerged : 0;
  if (split_groups) *split_groups = d ? d->nSplit : 0;
}

}  // extern "C"
