// fem_run -- command-line driver over the C-ABI (include/fem_hip.h, fem_dist.h; no HIP header, no Python).
//
// The caller side of the hot path, as the reference has it in src/main.cpp:
//   main            (main.cpp:116-154)  mesh file [-d], loop over partition counts, "Best time ..."
//   initializeData  (main.cpp:169-192)  partition -> initMeshStats -> IC -> dt = 0.1 maxEdgeLen^2
//   initializeGPUData (main.cpp:207-342) -> fem_ref_layout_create
//   GPUrun, non-visual path (main.cpp:344-617): limits, uploads, two warm-up `step`s, the check
//                   against CPUrun, then maxIt x 2 back-to-back `step`s between two time stamps
//   CPUrun          (main.cpp:619-676)  host steps from the three compute*CPU operators
// What differs, on purpose: the check uses the kernel's constants (the reference's host constants
// differ from its kernel's, so its own error counts mean nothing) and compares round 2 with TWO host
// steps (the reference restarts CPUrun from the initial state); the IC is seeded; --native runs the
// same protocol on the native fp32 plan; --out writes the colour pattern (ResultantPattern.off's shape).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fem_dist.h"
#include "../../include/fem_hip.h"

namespace {

struct Options {
  std::string file, out;
  bool debug = false, verbose = false, native = false, hostConstants = false;
  int startN = 52, endN = 52;        // main.cpp:100-101
  int maxIt = 1000;                  // main.cpp:548
  int fuse = 6, rowsPerPart = 820;   // native plan: fused steps per launch, rows per partition
  unsigned seed = 1;
  // --ranks N: the native plan sharded over N ranks (one host thread and, where there are that many, one GPU each),
  // halo exchange over RCCL (or device-to-device copies inside this process: --transport inproc)
  int ranks = 0, krings = 0, precision = FEM_PREC_F32;
  int imex = 0;                      // --imex ITERS: semi-implicit steps, ITERS CG iterations each (sharded mode)
  double imexFused = 0;              // --imex-fused TOL: semi-implicit steps on the shards' fused tiles, Chebyshev iteration to TOL (sharded mode)
  double dtScale = 1.0;              // --dt-scale: time step = scale x 0.1 maxEdgeLen^2
  double watchdog = -1;              // --watchdog SECONDS (sharded runs; default 600, 0 = off): a run that has not finished by then reports where every rank is and exits 124
  bool inproc = false;
  bool mixed = false;                // --mixed: fp64 plans in the reference's own mix (fp32 gradient branch), fused sharded runs
  std::string dump;                  // raw doubles: n[V] then c[V] of the final state, ordered numbering
};

struct Run {                         // what initializeData leaves behind
  fem_mesh *ordered = nullptr;
  std::vector<fem_uint> nodeParts, elementParts;
  std::vector<double> nFn, cFn;
  double dt = 0;
  int V = 0, F = 0;
  ~Run() { fem_mesh_destroy(ordered); }
};

bool ok(int rc, const char *what) {
  if (rc >= 0) return true;
  std::fprintf(stderr, "%s: %s (%s)\n", what, fem_error_string(rc), fem_error_detail());
  return false;
}

bool endsWith(const std::string &s, const char *suffix) {
  const size_t n = std::strlen(suffix);
  return s.size() >= n && !strcasecmp(s.c_str() + s.size() - n, suffix);
}

fem_mesh *loadMesh(const Options &o) {
  int V = 0, F = 0;
  double *pts = nullptr;
  fem_uint *tri = nullptr;
  unsigned char *rgb = nullptr;
  const int rc = endsWith(o.file, ".ply") ? fem_ply_read(o.file.c_str(), &V, &F, &pts, &tri, &rgb)
                                          : fem_off_read(o.file.c_str(), &V, &F, &pts, &tri, &rgb);
  if (!ok(rc, o.file.c_str())) return nullptr;
  fem_mesh *m = nullptr;
  const bool made = ok(fem_mesh_create(V, F, pts, tri, &m), "mesh");
  fem_free(pts); fem_free(tri); fem_free(rgb);
  return made ? m : nullptr;
}

bool initializeData(const fem_mesh *original, int n, const Options &o, const fem_model &model, Run &r) {
  fem_mesh_counts(original, &r.V, &r.F, nullptr);
  std::vector<int64_t> permV(r.V), permF(r.F);
  r.nodeParts.resize(n + 1); r.elementParts.resize(n + 2);
  if (!ok(fem_partition(original, n, permV.data(), permF.data(), r.nodeParts.data(), r.elementParts.data(), &r.ordered), "partition"))
    return false;
  if (!ok(fem_mesh_init_stats(r.ordered, o.native ? FEM_PREC_F64 : FEM_PREC_MIXED), "initMeshStats")) return false;
  double maxEdge = 0;
  if (!ok(fem_mesh_stats(r.ordered, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &maxEdge, nullptr, nullptr), "stats")) return false;
  r.nFn.resize(r.V); r.cFn.resize(r.V);
  fem_initial_condition(r.V, o.seed, model.nMax, r.nFn.data(), r.cFn.data());
  r.dt = o.dtScale * 0.1 * maxEdge * maxEdge;
  return true;
}

// CPUrun: `steps` host steps from the initial state, into n/c
bool hostRun(const Run &r, const fem_model &m, int steps, std::vector<double> &n, std::vector<double> &c) {
  n = r.nFn; c = r.cFn;
  std::vector<double> lapN(r.V), lapC(r.V), faceG(3 * (size_t)r.F), gradN(3 * (size_t)r.V), gradC(3 * (size_t)r.V);
  for (int s = 0; s < steps; s++) {
    if (fem_compute_laplacian_cpu(r.ordered, n.data(), lapN.data()) < 0 ||
        fem_compute_laplacian_cpu(r.ordered, c.data(), lapC.data()) < 0 ||
        fem_compute_face_gradients_cpu(r.ordered, n.data(), faceG.data()) < 0 ||
        fem_compute_vertex_gradients_cpu(r.ordered, faceG.data(), gradN.data()) < 0 ||
        fem_compute_face_gradients_cpu(r.ordered, c.data(), faceG.data()) < 0 ||
        fem_compute_vertex_gradients_cpu(r.ordered, faceG.data(), gradC.data()) < 0)
      return ok(fem_last_error(), "CPUrun");
    for (int j = 0; j < r.V; j++) {
      const float dot = (float)gradC[3 * j] * (float)gradN[3 * j] + (float)gradC[3 * j + 1] * (float)gradN[3 * j + 1] +
                        (float)gradC[3 * j + 2] * (float)gradN[3 * j + 2];
      const double dn = m.D * lapN[j] - m.alpha * n[j] * lapC[j] - m.alpha * dot + m.S * m.r * n[j] * (m.nMax - n[j]);
      const double dc = lapC[j] + m.S * (-c[j] + n[j] / (1 + n[j]));
      n[j] = std::max(0.0, n[j] + r.dt * dn);
      c[j] = std::max(0.0, c[j] + r.dt * dc);
    }
  }
  return true;
}

int reportRound(int round, const std::vector<double> &gn, const std::vector<double> &gc, const std::vector<double> &hn,
                const std::vector<double> &hc, bool print) {
  double sumN = 0, sumC = 0;
  int errors = 0;
  const int V = (int)gn.size();
  for (int i = 0; i < V; i++) {
    if (std::isnan(hn[i])) continue;
    const double en = std::fabs(gn[i] - hn[i]), ec = std::fabs(gc[i] - hc[i]);
    sumN += en; sumC += ec;
    if (en > 0 || ec > 0) errors++;
  }
  if (print)
    std::printf("Round %d: Sum / Mean error for n: %g / %g and for c: %g / %g. Number of errors in total: %d(%d)\n", round,
                sumN, sumN / V, sumC, sumC / V, errors, V);
  return errors;
}

void writePattern(const Options &o, const Run &r, const std::vector<double> &n, const std::vector<double> &c) {
  // format(): min/max + colour map on the device (fem_plan_format), then the COFF file
  fem_plan *plan = nullptr;
  if (!ok(fem_plan_create(r.ordered, FEM_PREC_F64, nullptr, &plan), "format plan")) return;
  std::vector<unsigned char> rgb(3 * (size_t)r.V);
  double lo = 0, hi = 0;
  const double *pts = nullptr;
  const fem_uint *tri = nullptr;
  fem_mesh_arrays(r.ordered, &pts, &tri);
  if (ok(fem_plan_set_state(plan, n.data(), c.data()), "format") &&
      ok(fem_plan_format(plan, (fem_uint)r.V, nullptr, rgb.data(), &lo, &hi), "format") &&
      ok(fem_off_write(o.out.c_str(), r.V, r.F, pts, tri, rgb.data()), o.out.c_str()))
    std::printf("Pattern written to %s (n in [%g, %g])\n", o.out.c_str(), lo, hi);
  fem_plan_destroy(plan);
}

double seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// GPUrun, non-visual: returns microseconds per step, -1 on a limit or error
double gpuRun(int n, const Options &o, const fem_model &model, const Run &r) {
  fem_ref_layout *lay = nullptr;
  fem_ref_device *dev = nullptr;
  if (!ok(fem_ref_layout_create(r.ordered, n, r.nodeParts.data(), r.elementParts.data(), &lay), "initializeGPUData")) return -1;
  const int rc = fem_ref_device_create(lay, r.nodeParts.data(), r.V, &dev);
  fem_ref_layout_destroy(lay);
  if (rc == FEM_ERR_LIMIT) { std::printf("%s\n", fem_error_detail()); return -2; }      // main.cpp:357-364: the loop goes on
  if (!ok(rc, "GPUrun")) return -1;

  std::vector<fem_double2> fn(r.V);
  for (int i = 0; i < r.V; i++) { fn[i].x = r.nFn[i]; fn[i].y = r.cFn[i]; }
  double usPerStep = -1;
  uint64_t one, two, fv, fvw, nbr, vtxW, vw, grads, parts, bfc;
  fem_uint fvPitch, fvwPitch, vwPitch, vvPitch, vwsPitch, vvSize, hePitch, blocks, threads, smem;
  fem_ref_device_args(dev, &one, &two, &fv, &fvw, &fvPitch, &fvwPitch, &nbr, &vtxW, &vwPitch, &vw, &vvPitch, &vwsPitch,
                      &vvSize, &grads, &hePitch, &parts, &bfc, &blocks, &threads, &smem);
  auto doStep = [&](uint64_t src, uint64_t dst) {
    step((fem_double2 *)src, (fem_double2 *)dst, (fem_uint *)fv, (double *)fvw, fvPitch, fvwPitch, (fem_uint *)nbr,
         (double *)vtxW, vwPitch, (double *)vw, vvPitch, vwsPitch, vvSize, (fem_float4 *)grads, hePitch, (fem_uint *)parts,
         (fem_uint *)bfc, blocks, threads, r.dt, smem);
  };
  auto download = [&](uint64_t src, std::vector<double> &gn, std::vector<double> &gc) {
    if (!ok(fem_device_download(fn.data(), src, sizeof(fem_double2) * (size_t)r.V), "read-back")) return false;
    gn.resize(r.V); gc.resize(r.V);
    for (int i = 0; i < r.V; i++) { gn[i] = fn[i].x; gc[i] = fn[i].y; }
    return true;
  };
  do {
    if (!ok(fem_ref_device_set_state(dev, fn.data()), "upload")) break;
    doStep(one, two);
    doStep(two, one);
    if (!ok(fem_device_synchronize(), "step") || !ok(fem_last_error(), "step")) break;

    // Correctness test (main.cpp:492-541)
    std::vector<double> gn, gc, hn, hc;
    int errors = 0;
    if (!hostRun(r, model, 1, hn, hc) || !download(two, gn, gc)) break;
    reportRound(1, gn, gc, hn, hc, o.verbose || o.debug);
    if (!hostRun(r, model, 2, hn, hc) || !download(one, gn, gc)) break;
    errors = reportRound(2, gn, gc, hn, hc, o.verbose || o.debug);
    if (o.debug) { usPerStep = 0; break; }

    // Speed test (main.cpp:547-575): maxIt x 2 launches, nothing in between
    const double t0 = seconds();
    for (int i = 0; i < o.maxIt; i++) { doStep(one, two); doStep(two, one); }
    if (!ok(fem_device_synchronize(), "step")) break;
    const double ms = (seconds() - t0) * 1e3 / 2;
    usPerStep = 1000 * ms / o.maxIt;
    if (o.verbose) {
      std::printf("Time needed for %d iterations: %f ms with %u threads each and %d partitions\n", o.maxIt, ms, threads, n);
      std::printf("Average time needed per Iteration: %f us\n", usPerStep);
    } else {
      std::printf("Errors: %d, Partitions: %d, Threads: %u, Time: %f\n", errors, n, threads, usPerStep);
    }
    if (!o.out.empty() && download(one, gn, gc)) writePattern(o, r, gn, gc);
  } while (false);
  fem_ref_device_destroy(dev);
  return usPerStep;
}

// the same protocol on the native plan (fused tiles; fuse = 0: sliced ELL): fp32, --f64, or --mixed (fp64 in the reference's
// own precision mix, fused tiles only)
double nativeRun(int n, const Options &o, const fem_model &model, const Run &r) {
  fem_plan *plan = nullptr;
  const int rc = o.fuse > 0 ? fem_plan_create_ex(r.ordered, n, r.nodeParts.data(), nullptr, o.precision, &model, o.fuse,
                                                 o.mixed ? FEM_PLAN_MIXED_RECORDS : 0u, &plan)
                            : fem_plan_create(r.ordered, o.precision, &model, &plan);
  if (rc == FEM_ERR_LIMIT) { std::printf("%s\n", fem_error_detail()); return -2; }
  if (!ok(rc, "plan")) return -1;
  double usPerStep = -1;
  do {
    std::vector<double> gn(r.V), gc(r.V), hn, hc;
    if (!ok(fem_plan_set_state(plan, r.nFn.data(), r.cFn.data()), "upload")) break;
    if (!ok(fem_plan_run(plan, 2, r.dt, &model, nullptr), "step") || !ok(fem_plan_get_state(plan, gn.data(), gc.data()), "read-back")) break;
    if (!hostRun(r, model, 2, hn, hc)) break;
    const int errors = reportRound(2, gn, gc, hn, hc, o.verbose || o.debug);
    if (o.debug) { usPerStep = 0; break; }
    const double t0 = seconds();
    if (!ok(fem_plan_run(plan, 2 * o.maxIt, r.dt, &model, nullptr), "step") || !ok(fem_device_synchronize(), "step")) break;
    const double ms = (seconds() - t0) * 1e3 / 2;
    usPerStep = 1000 * ms / o.maxIt;
    int threads = 0;
    fem_plan_tile_info(plan, nullptr, nullptr, nullptr, nullptr, &threads, nullptr);
    if (o.verbose) {
      std::printf("Time needed for %d iterations: %f ms with %d threads each and %d partitions\n", o.maxIt, ms, threads, n);
      std::printf("Average time needed per Iteration: %f us\n", usPerStep);
    } else {
      std::printf("Errors: %d, Partitions: %d, Threads: %d, Time: %f\n", errors, n, threads, usPerStep);
    }
    if (!o.out.empty() && ok(fem_plan_get_state(plan, gn.data(), gc.data()), "read-back")) writePattern(o, r, gn, gc);
  } while (false);
  fem_plan_destroy(plan);
  return usPerStep;
}

class ThreadBarrier {
 public:
  explicit ThreadBarrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lock(m_);
    const int gen = gen_;
    if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); }
    else cv_.wait(lock, [&] { return gen != gen_; });
  }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, count_ = 0, gen_ = 0;
};

// The native protocol (two steps, check, 2 x maxIt timed steps) with the mesh sharded along the partition ranges
// over o.ranks ranks: every rank steps its shard and exchanges boundary state every krings steps (fem_dist_run).
double shardedRun(int n, const Options &o, const fem_model &model, const Run &r) {
  const int R = o.ranks;
  if (n < R) { std::fprintf(stderr, "--ranks %d needs at least that many partitions (have %d)\n", R, n); return -1; }
  const int krings = o.krings > 0 ? o.krings : (o.imex > 0 ? 1 : (o.imexFused > 0 ? std::max(1, o.fuse) : std::max(2, 2 * o.fuse)));
  double wsumMax = 0;                // the Chebyshev solver's interval: half the whole mesh's spectrum bound (the same on every rank)
  if (o.imexFused > 0) {
    if (!ok(fem_mesh_spectrum_bound(r.ordered, &wsumMax), "spectrum bound") || !(wsumMax > 0)) return -1;
    wsumMax *= 0.5;
  }
  const int ndev = std::max(1, fem_dist_device_count());
  if (!o.inproc && R > ndev) {     // RCCL refuses (or hangs on) two ranks of one communicator on the same GPU
    std::fprintf(stderr, "--ranks %d over RCCL needs that many GPUs (this host shows %d); --transport inproc lets ranks share one\n", R, ndev);
    return -1;
  }
  char uid[128] = {0};
  fem_dist_world *world = nullptr;
  if (o.inproc) { if (!ok(fem_dist_world_create(R, &world), "world")) return -1; }
  else if (fem_dist_rccl_unique_id(uid) != FEM_OK) { std::fprintf(stderr, "RCCL: %s\n", fem_dist_error_detail()); return -1; }
  std::vector<double> gn(r.V), gc(r.V);
  std::atomic<int> failed{0};
  ThreadBarrier bar(R);
  std::vector<double> rankMs(R, 0.0);
  std::vector<int> exchanges(R, 0), mergedG(R, 0), splitG(R, 0);
  std::vector<fem_uint> ghosts(R, 0);
  int errorsRound2 = 0;
  std::vector<double> hn, hc;
  const bool check = r.V <= 4000000 && o.imex == 0 && !(o.imexFused > 0);   // CPUrun (explicit steps) of a large mesh takes minutes: check up to 4M vertices
  if (check && !hostRun(r, model, 2, hn, hc)) return -1;

  // where every rank is, for the watchdog: a run that hangs (a rank waiting for the GPU or for another rank) ends with
  // a message instead of at the caller's time limit
  std::vector<std::atomic<const char *>> stage(R);
  for (auto &s : stage) s = "start";
  std::mutex wdMutex;
  std::condition_variable wdCv;
  bool wdDone = false;
  std::thread watchdog;
  // sharded runs always have one (default 600 s, --watchdog 0 turns it off): a rank stuck in a collective or waiting
  // for the GPU must end the process with a message and a non-zero status, not hang its caller
  const double wdSeconds = o.watchdog < 0 ? 600.0 : o.watchdog;
  if (wdSeconds > 0)
    watchdog = std::thread([&]() {
      std::unique_lock<std::mutex> lock(wdMutex);
      if (wdCv.wait_for(lock, std::chrono::duration<double>(wdSeconds), [&] { return wdDone; })) return;
      for (int i = 0; i < R; i++) std::fprintf(stderr, "watchdog: rank %d is at: %s\n", i, stage[i].load());
      std::fflush(stderr);
      std::_Exit(124);
    });

  auto body = [&](int rank) {
    fem_shard *shard = nullptr;
    fem_plan *plan = nullptr;
    fem_dist *dist = nullptr;
    void *comm = nullptr, *stream = nullptr;
    auto bail = [&](const char *what, bool distError) {
      std::fprintf(stderr, "rank %d: %s: %s\n", rank, what, distError ? fem_dist_error_detail() : fem_error_detail());
      failed = 1;
    };
    bool up = false;
    do {
      stage[rank] = "set-up: device";
      if (fem_dist_set_device(rank % ndev) != FEM_OK) { bail("device", true); break; }
      stage[rank] = "set-up: shard";
      if (fem_shard_create(r.ordered, n, r.nodeParts.data(), rank, R, krings, &shard) != FEM_OK) { bail("shard", false); break; }
      stage[rank] = "set-up: plan";
      const int rc = o.imex > 0 ? fem_plan_create_ex(r.ordered, 0, nullptr, shard, o.precision, &model, 0, FEM_PLAN_IMPLICIT, &plan)
                     : (o.fuse > 0 ? fem_plan_create_ex(r.ordered, n, r.nodeParts.data(), shard, o.precision, &model, o.fuse,
                                                        o.mixed ? FEM_PLAN_MIXED_RECORDS : 0u, &plan)
                                   : fem_plan_create_shard(r.ordered, shard, o.precision, &model, &plan));
      if (rc != FEM_OK) { bail("plan", false); break; }
      fem_uint ownB = 0, ownE = 0, nLocal = 0, nSend = 0;
      fem_shard_info(shard, &ownB, &ownE, &nLocal, &nSend);
      ghosts[rank] = nLocal - (ownE - ownB);
      const fem_uint *l2g = nullptr;
      fem_shard_arrays(shard, &l2g, nullptr, nullptr, nullptr);
      std::vector<double> ln(nLocal), lc(nLocal);
      for (fem_uint i = 0; i < nLocal; i++) { ln[i] = r.nFn[l2g[i]]; lc[i] = r.cFn[l2g[i]]; }
      stage[rank] = "set-up: upload";
      if (fem_plan_set_state(plan, ln.data(), lc.data()) != FEM_OK) { bail("upload", false); break; }
      stage[rank] = "set-up: stream";
      if (fem_dist_stream_create(&stream) != FEM_OK) { bail("stream", true); break; }
      up = true;
    } while (false);
    // ncclCommInitRank is a collective: a rank that failed above would leave the others waiting inside it for ever,
    // so all ranks meet first and nobody creates a communicator unless everybody got this far
    stage[rank] = "local set-up done, waiting for the other ranks";
    bar.wait();
    if (failed) up = false;
    if (up) {
      stage[rank] = "set-up: communicator";
      if (!o.inproc && fem_dist_rccl_comm_create(uid, rank, R, &comm) != FEM_OK) { bail("RCCL communicator", true); up = false; }
      else if (fem_dist_create(plan, shard, comm, world, &dist) != FEM_OK) { bail("fem_dist_create", true); up = false; }
    }
    stage[rank] = "set-up done, waiting for the other ranks";
    bar.wait();                                        // every rank is set up (or has failed)
    if (failed) up = false;
    auto download = [&]() {
      fem_uint ownB = 0, ownE = 0, nLocal = 0;
      fem_shard_info(shard, &ownB, &ownE, &nLocal, nullptr);
      std::vector<double> ln(nLocal), lc(nLocal);
      stage[rank] = "download (device synchronize)";
      if (fem_dist_wait(dist, stream) != FEM_OK || fem_plan_get_state(plan, ln.data(), lc.data()) != FEM_OK) { failed = 1; return; }
      std::copy(ln.begin(), ln.begin() + (ownE - ownB), gn.begin() + ownB);
      std::copy(lc.begin(), lc.begin() + (ownE - ownB), gc.begin() + ownB);
    };
    if (up) {
      // two warm-up steps and the check against CPUrun (main.cpp:480-541)
      stage[rank] = "two warm-up steps (issuing)";
      if ((o.imexFused > 0 ? fem_dist_run_imex_fused(dist, 2, r.dt, &model, o.imexFused, wsumMax, stream)
                           : (o.imex > 0 ? fem_dist_run_imex(dist, 2, r.dt, &model, o.imex, stream) : fem_dist_run(dist, 2, r.dt, &model, stream))) != FEM_OK) { bail("step", true); }
      else download();
    }
    stage[rank] = "warm-up done, waiting for the other ranks";
    bar.wait();
    if (rank == 0 && !failed && check) errorsRound2 = reportRound(2, gn, gc, hn, hc, o.verbose || o.debug);
    bar.wait();
    if (up && !failed && !o.debug) {
      const double t0 = seconds();
      stage[rank] = "timed steps (issuing)";
      if ((o.imexFused > 0 ? fem_dist_run_imex_fused(dist, 2 * o.maxIt, r.dt, &model, o.imexFused, wsumMax, stream)
                           : (o.imex > 0 ? fem_dist_run_imex(dist, 2 * o.maxIt, r.dt, &model, o.imex, stream) : fem_dist_run(dist, 2 * o.maxIt, r.dt, &model, stream))) != FEM_OK ||
          fem_dist_wait(dist, stream) != FEM_OK ||
          (stage[rank] = "timed steps (stream synchronize)", fem_dist_stream_synchronize(stream)) != FEM_OK)
        bail("step", true);
      rankMs[rank] = (seconds() - t0) * 1e3;
      fem_dist_stats(dist, &exchanges[rank], &mergedG[rank], &splitG[rank]);
      if (!failed && (!o.dump.empty() || !o.out.empty())) download();
    }
    stage[rank] = "timed steps done, waiting for the other ranks";
    bar.wait();
    stage[rank] = "teardown";
    fem_dist_destroy(dist);
    fem_dist_rccl_comm_destroy(comm);
    fem_dist_stream_destroy(stream);
    fem_plan_destroy(plan);
    fem_shard_destroy(shard);
    stage[rank] = "finished";
  };
  std::vector<std::thread> threads;
  for (int rank = 0; rank < R; rank++) threads.emplace_back(body, rank);
  for (auto &t : threads) t.join();
  if (watchdog.joinable()) {
    { std::lock_guard<std::mutex> lock(wdMutex); wdDone = true; }
    wdCv.notify_all();
    watchdog.join();
  }
  fem_dist_world_destroy(world);
  if (failed) return -1;
  if (o.debug) return 0;
  const double ms = *std::max_element(rankMs.begin(), rankMs.end()) / 2;     // slowest rank, per maxIt iterations of two steps
  const double usPerStep = 1000 * ms / o.maxIt;
  fem_uint ghostMax = *std::max_element(ghosts.begin(), ghosts.end());
  std::printf("Ranks: %d (%s, %d device%s), ghost rings: %d, exchanges: %d, groups merged/split: %d/%d, ghost rows (max): %u\n", R,
              o.inproc ? "in-process copies" : "RCCL", ndev, ndev == 1 ? "" : "s", krings, exchanges[0], mergedG[0], splitG[0], ghostMax);
  if (o.verbose) {
    std::printf("Time needed for %d iterations: %f ms and %d partitions\n", o.maxIt, ms, n);
    std::printf("Average time needed per Iteration: %f us\n", usPerStep);
  } else {
    std::printf("Errors: %d, Partitions: %d, Threads: %d, Time: %f\n", errorsRound2, n, R, usPerStep);
  }
  if (!o.dump.empty()) {
    FILE *f = std::fopen(o.dump.c_str(), "wb");
    if (!f || std::fwrite(gn.data(), 8, gn.size(), f) != gn.size() || std::fwrite(gc.data(), 8, gc.size(), f) != gc.size())
      std::fprintf(stderr, "cannot write %s\n", o.dump.c_str());
    if (f) std::fclose(f);
  }
  if (!o.out.empty()) writePattern(o, r, gn, gc);
  return usPerStep;
}

int usage() {
  std::fprintf(stderr,
               "usage: fem_run MESH.off|MESH.ply [-d] [-v] [-n PARTS | -n FIRST:LAST] [--iterations N] [--seed S]\n"
               "               [--native [--fuse K] [--rows R]] [--host-constants] [--out PATTERN.off]\n"
               "  -d               check only (two steps against the host operators), as the reference's -d\n"
               "  -v               the reference's verbose prints\n"
               "  -n               partition counts to try (default 52, the reference's start_n..end_n)\n"
               "  --iterations     timed iterations of two steps each (default 1000)\n"
               "  --native         native fp32 plan instead of step() on the reference layout; partitions of R rows\n"
               "                   (default 820) unless -n is given, K fused steps per launch (default 6, 0 = one)\n"
               "  --host-constants r = 1.52, alpha = 12.02 (main.cpp:49-54) instead of the kernel's set; native only\n"
               "  --out            write the coloured result (COFF)\n"
               "  --ranks N        native plan sharded over N ranks (threads; GPU = rank mod device count), halo exchange over\n"
               "                   RCCL every --rings K steps (default 2 x fuse); --transport inproc: device-to-device copies\n"
               "                   inside this process instead (all ranks may share one GPU); --f64: fp64 plans;\n"
               "                   --mixed: fp64 plans in the reference's own precision mix (fp32 gradient branch; fused runs);\n"
               "                   (--f64 and --mixed also without --ranks: the native single-GPU plan in that precision)\n"
               "                   --dump FILE: final state as raw doubles n[V], c[V] (ordered numbering);\n"
               "                   --imex-fused TOL: semi-implicit steps on the fused tiles (Chebyshev iteration to relative tolerance\n"
               "                   TOL, --fuse K iterations per launch and between halo exchanges; no all-reduce);\n"
               "                   --imex ITERS: semi-implicit steps (diffusion by ITERS CG iterations, dot products all-reduced\n"
               "                   over the ranks), --dt-scale S: time step = S x 0.1 maxEdgeLen^2;\n"
               "                   --watchdog SECONDS (default 600, 0 = off): if the run has not finished by then, say where every rank\n"
               "                   is and exit 124\n"
               "                   (FEM_DIST_TRACE=1: one stderr line per exchange stage)\n");
  return 2;
}

}  // namespace

int main(int argc, char **argv) {
  Options o;
  bool partsGiven = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto value = [&]() -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
    if (a == "-d") o.debug = true;
    else if (a == "-v") o.verbose = true;
    else if (a == "--native") o.native = true;
    else if (a == "--host-constants") o.hostConstants = true;
    else if (a == "--f64") { o.precision = FEM_PREC_F64; o.native = true; }
    else if (a == "--mixed") { o.precision = FEM_PREC_F64; o.mixed = true; o.native = true; }
    else if (a == "-n" || a == "--iterations" || a == "--seed" || a == "--fuse" || a == "--rows" || a == "--out" ||
             a == "--ranks" || a == "--rings" || a == "--transport" || a == "--dump" || a == "--imex" || a == "--imex-fused" || a == "--dt-scale" || a == "--watchdog") {
      const char *v = value();
      if (!v) return usage();
      if (a == "-n") {
        partsGiven = true;
        o.startN = o.endN = std::atoi(v);
        if (const char *colon = std::strchr(v, ':')) o.endN = std::atoi(colon + 1);
      } else if (a == "--iterations") o.maxIt = std::atoi(v);
      else if (a == "--seed") o.seed = (unsigned)std::strtoul(v, nullptr, 10);
      else if (a == "--fuse") o.fuse = std::atoi(v);
      else if (a == "--rows") o.rowsPerPart = std::atoi(v);
      else if (a == "--ranks") { o.ranks = std::atoi(v); o.native = true; }
      else if (a == "--rings") o.krings = std::atoi(v);
      else if (a == "--transport") { if (!std::strcmp(v, "inproc")) o.inproc = true; else if (std::strcmp(v, "rccl")) return usage(); }
      else if (a == "--dump") o.dump = v;
      else if (a == "--imex") o.imex = std::atoi(v);
      else if (a == "--imex-fused") o.imexFused = std::atof(v);
      else if (a == "--dt-scale") o.dtScale = std::atof(v);
      else if (a == "--watchdog") o.watchdog = std::atof(v);
      else o.out = v;
    } else if (!a.empty() && a[0] == '-') return usage();
    else if (o.file.empty()) o.file = a;
    else return usage();
  }
  if (o.file.empty() || o.startN < 1 || o.endN < o.startN || o.maxIt < 1 || o.fuse < 0 || o.fuse > 8 || o.rowsPerPart < 1 ||
      (o.hostConstants && !o.native) || (o.mixed && (!o.native || o.fuse < 1 || o.imex > 0)) || (o.precision == FEM_PREC_F64 && !o.native) || o.ranks < 0 || o.ranks > 64 || o.krings < 0 || o.imex < 0 || (o.imex > 0 && o.ranks < 1) ||
      o.imexFused < 0 || o.imexFused >= 1 || (o.imexFused > 0 && (o.ranks < 1 || o.fuse < 1 || o.imex > 0)) ||
      !(o.dtScale > 0))
    return usage();

  fem_model model;
  fem_default_model(&model);
  if (o.hostConstants) { model.r = 1.52; model.alpha = 12.02; }

  fem_mesh *original = loadMesh(o);
  if (!original) return 1;
  if (o.native) {
    // irregular meshes on the native plans: rows of more than 6 entries numbered first, so that they lead every part
    // and share waves in the fused kernel (the Python drivers do the same: fem_amd/dist.py:_partition)
    fem_mesh *renumbered = nullptr;
    const int wide = fem_mesh_wide_rows_first(original, 6, nullptr, &renumbered);
    if (!ok(wide, "wide rows first")) { fem_mesh_destroy(original); return 1; }
    if (renumbered) { fem_mesh_destroy(original); original = renumbered; }
    if (o.verbose) std::printf("%d rows of more than 6 entries%s\n", wide, wide ? ": numbered first" : "");
  }
  int V = 0;
  fem_mesh_counts(original, &V, nullptr, nullptr);
  if (o.native && !partsGiven) o.startN = o.endN = std::max(1, V / o.rowsPerPart);

  double bestTime = 0;
  int bestN = -1, failedRuns = 0;
  for (int n = o.startN; n <= o.endN; n++) {
    Run r;
    if (!initializeData(original, n, o, model, r)) { fem_mesh_destroy(original); return 1; }
    const double t = o.ranks > 0 ? shardedRun(n, o, model, r) : (o.native ? nativeRun(n, o, model, r) : gpuRun(n, o, model, r));
    if ((bestN == -1 || bestTime > t) && t > 0) { bestN = n; bestTime = t; }
    if (t < 0 && t > -1.5) failedRuns++;       // -2: this partition count exceeds the size limits (as the reference, not a failure)
  }
  if (!o.debug) std::printf("Best time with %d partitions is %.1f!\n", bestN, bestTime);      // also with bestN = -1, as the reference
  fem_mesh_destroy(original);
  return failedRuns ? 1 : 0;                 // a run that failed has said why on stderr
}
