// device.hip -- the device half of the C-ABI: plans, the reference device harness, launches.
// The gfx950 kernels live in kernels_*.hpp (same translation unit):
//   kernels_step.hpp     ref_step_kernel (the reference's fused time step on its own padded SoA
//                        layout, what `step` launches; src/FEM_common.cu:82-180),
//                        sell_step_kernel (sliced ELL, one wave64 per 64-row slice),
//                        tile_step_kernel (k fused steps per launch, rows in registers, state in LDS)
//   kernels_assembly.hpp operator assembly from the vertex positions, per-face records, deformation
//   kernels_output.hpp   min/max + colour map (format()), halo pack
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "fem_host.hpp"

namespace fem {
thread_local int g_last_error = FEM_OK;
thread_local std::string g_error_detail;
int fail(int status, const char *what) {
  g_last_error = status;
  g_error_detail = what ? what : "";
  return status;
}
}  // namespace fem

namespace {

int fail(int status, const char *what, hipError_t e = hipSuccess) {
  fem::fail(status, what);
  if (e != hipSuccess) { fem::g_error_detail += ": "; fem::g_error_detail += hipGetErrorString(e); }
  return status;
}

#define HIP_TRY(call)                                          \
  do {                                                         \
    hipError_t e_ = (call);                                    \
    if (e_ != hipSuccess) return fail(FEM_ERR_DEVICE, #call, e_); \
  } while (0)

struct ModelD { double nMax, D, r, alpha, S; };

#include "kernels_common.hpp"
#include "kernels_step.hpp"
#include "kernels_assembly.hpp"
#include "kernels_output.hpp"
#include "kernels_implicit.hpp"

template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    n = count;
    return hipMalloc((void **)&p, sizeof(T) * (count ? count : 1));
  }
  hipError_t upload(const T *h) { return hipMemcpy(p, h, sizeof(T) * n, hipMemcpyHostToDevice); }
  void release() { if (p) (void)hipFree(p); p = nullptr; }
};

size_t pitchFor(size_t widthBytes) { return (widthBytes + 511) / 512 * 512; }   // cudaMallocPitch-like

}  // namespace

// ===========================================================================
// C-ABI, device half
// ===========================================================================

// Register-tile geometries of the fused kernel: threads per workgroup, rows per thread, minimum
// waves per SIMD (sets the VGPR budget: 4 -> 128, 3 -> 168).  FEM_TILE_CONFIG picks one (tuning knob); the
// defaults below are the measured best: 8 = 512 x 4 (2048 register-resident rows per tile, 128 VGPRs, 2 workgroups
// per CU) for large meshes without rows wider than 6 entries, 2 = 512 x 3 otherwise.
constexpr int kTileW = 6;
constexpr fem_uint kTileMaxLocal = 4096;
constexpr fem_uint kTileMaxExtra = 1536;  // LDS-resident overflow entries per tile (rows with more than kTileW neighbours)
struct TileCfg { int tpb, r, minw; };
constexpr TileCfg kTileCfgs[] = {{512, 2, 4}, {256, 4, 3}, {512, 3, 4}, {256, 3, 4}, {256, 2, 4}, {1024, 2, 4}, {512, 2, 6}, {768, 2, 6}, {512, 4, 4}};
constexpr int kNumTileCfgs = sizeof(kTileCfgs) / sizeof(kTileCfgs[0]);
constexpr int kDefaultTileCfg = 8;       // 512 threads x 4 register rows: 2048-row tiles, 128 VGPRs, 2 workgroups per CU
constexpr fem_uint kLargeMeshRows = 1500000;
constexpr int kDefaultTileCfgWide = 2;   // 512 x 3: 1536-row tiles, 120 VGPRs (meshes with rows of more than 6 entries)


struct fem_ref_device {
  int n = 0, V = 0;
  fem_uint maxV = 0, W = 0, maxF = 0, threads = 0, smem = 0;
  size_t vw_pitch = 0, nbr_pitch = 0, vvw_pitch = 0, he_pitch = 0, fv_pitch = 0, fvw_pitch = 0;
  double2 *fn_one = nullptr, *fn_two = nullptr;
  double *vtxW = nullptr, *vv_weights = nullptr, *fv_weights = nullptr;
  fem_uint *nbr = nullptr, *fv = nullptr, *parts_n = nullptr, *bfc = nullptr;
  float4 *he_grads = nullptr;
  bool one_is_current = true;
};

struct fem_plan {
  int precision = FEM_PREC_F32;
  fem_uint V = 0, nslices = 0, nblocks = 0;
  size_t entries = 0;
  DevBuf<fem_uint> slice_off, idx, send_idx;
  // fused-tile path (ksteps > 0): replaces the sliced-ELL arrays
  bool mixed = false;               // FEM_PLAN_MIXED_RECORDS: fp64 weights + fp32 gradient coefficients in the fused tiles
  int ksteps = 0, tcfg = 0;
  // tile_step_persist_kernel: up to `persist` groups of k steps per launch (0: one launch per group; FEM_TILE_PERSIST,
  // else by tile count: persistRule); its control words, the per-tile "groups finished" flags, the tiles each tile waits for
  int persist = -1, persist_min = 2, persist_deps = 1, persist_grid = 0;
  DevBuf<fem_uint> pctl, pdepOff, pdep;
  int device = -1;                  // ordinal of the GPU that was current when the plan was created (owns every buffer)
  int prologue = -1, sell_nt = -1;   // FEM_TILE_PROLOGUE / FEM_SELL_NT as read when the plan was created (-1: by operator size)
  fem_uint ntiles = 0, maxLocal = 0, maxExtra = 0;
  fem_uint nInterior = 0, nBoundary = 0, nGhost = 0;   // tile classes of a shard's plan, in this order
  uint64_t tile_rows = 0, tile_entries = 0;
  uint64_t tile_bytes = 0;
  DevBuf<fem_uint> tmeta, tghost, tidx2, txinfo;
  void *tcoef = nullptr;
  bool packed = false;                 // fp32 plans: 12-byte operator records (kernels_step.hpp, PK)
  float *rscale = nullptr;             // ... of a sliced-ELL plan: the row scales (a fused plan keeps them inside its tile blocks)
  // device assembly (FEM_PLAN_DEVICE_ASSEMBLY): connectivity SoA, positions, per-face records, entry tables
  bool can_assemble = false;
  bool fan_rows = false;            // every row is one fan in order: assemble_row_fan applies
  bool wide_rows = false;           // some row has more than 6 entries
  fem_uint max_row = 0;             // entries of the longest row
  bool rows_are_vertices = false;   // row r is mesh vertex r (whole-mesh plan): a neighbour's mesh id is its row id
  fem_uint F = 0, Vg = 0;
  DevBuf<fem_uint> tri0, tri1, tri2, topo, rowg;
  void *pos = nullptr, *pos0 = nullptr, *rec = nullptr;      // T4[Vg], T4[Vg], T4[4F]
  double *maxedge = nullptr, *blockmax = nullptr;
  // semi-implicit stepping (FEM_PLAN_IMPLICIT): lumped mass and row sums of w per row; CG work vectors
  bool can_imex = false, imex_ready = false;
  // semi-implicit steps on the fused tiles (fem_plan_run_imex_fused): largest row sum of the Laplacian weights (spectrum
  // bound), the weights alone in tile order, right-hand side, second iterate buffer, the two direction buffers
  double wsum_max = 0;
  void *cb_tw = nullptr, *cb_b = nullptr, *cb_x = nullptr, *cb_d[2] = {nullptr, nullptr};
  int cheb_iters = 0;
  double imex_tol = 0;              // fem_plan_imex_tolerance: relative residual at which a solve stops (0: fixed iteration count)
  void *mass = nullptr, *wsum = nullptr, *ix_w = nullptr, *ix_r = nullptr, *ix_p = nullptr, *ix_q = nullptr;
  double *ix_partial = nullptr, *ix_sc = nullptr;
  void *coef = nullptr;            // float4 / double4 [entries]
  void *state[2] = {nullptr, nullptr};   // deviations from (base_n, base_c), interleaved
  int cur = 0;
  double base_n = 1.0, base_c = 0.5;
};

template <typename T> static int setPoints(fem_plan *p, const double *pts) {
  std::vector<T> h(4 * (size_t)p->Vg);
  for (size_t i = 0; i < p->Vg; i++) { h[4 * i] = (T)pts[3 * i]; h[4 * i + 1] = (T)pts[3 * i + 1]; h[4 * i + 2] = (T)pts[3 * i + 2]; h[4 * i + 3] = 0; }
  HIP_TRY(hipMemcpy(p->pos0, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->pos, p->pos0, sizeof(T) * h.size(), hipMemcpyDeviceToDevice));
  return FEM_OK;
}

template <typename T> static void launchAssemble(fem_plan *p, hipStream_t st) {
  typedef typename Vec<T>::t4 T4;
  // Fan-ordered plans: the rows of at most 6 entries go to a kernel of their own (84 VGPRs, 5 waves per
  // SIMD); wider rows, if the plan has any, to a second launch whose partial maxima land behind the first one's.
  fem_uint nb, parts = 1;
  if (p->ksteps > 0) {
    nb = p->ntiles;
    auto launch = [&](auto kern, double *bm) {
      hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, st, (const T4 *)p->pos, (const uint2 *)p->topo.p, p->tmeta.p, p->tghost.p,
                         p->txinfo.p, p->rowg.p, (T4 *)p->tcoef, (fem_uint)kTileW, bm);
    };
    if (!p->fan_rows) launch(assemble_rows_tile_kernel<T, false>, p->blockmax);
    else {
      launch(assemble_rows_tile_kernel<T, true, 1>, p->blockmax);
      if (p->wide_rows) { launch(assemble_rows_tile_kernel<T, true, 2>, p->blockmax + nb); parts = 2; }
    }
  } else {
    nb = p->nblocks;
    auto launch = [&](auto kern, double *bm) {
      hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, st, (const T4 *)p->pos, (const uint2 *)p->topo.p, p->slice_off.p, p->rowg.p,
                         (T4 *)p->coef, p->nslices, p->V, bm);
    };
    if (!p->fan_rows) launch(assemble_rows_sell_kernel<T, false>, p->blockmax);
    else {
      launch(assemble_rows_sell_kernel<T, true, 1>, p->blockmax);
      if (p->wide_rows) { launch(assemble_rows_sell_kernel<T, true, 2>, p->blockmax + nb); parts = 2; }
    }
  }
  hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(1024), 0, st, (const double *)p->blockmax, parts * nb, p->maxedge);
}

// Assembly from the current positions fused with one time step (config 5's iteration minus the deformation).
template <typename T> static void launchAssembleStep(fem_plan *p, double dt, const fem_model *model, bool store, hipStream_t st) {
  typedef typename Vec<T>::t2 T2;
  typedef typename Vec<T>::t4 T4;
  const ModelT<T> m = {(T)(model->nMax - p->base_n), (T)model->D, (T)model->r, (T)model->alpha, (T)model->S, (T)dt, (T)p->base_n,
                       (T)(1.0 / (1.0 + p->base_n))};
  auto pick = [&](auto ident) {
    constexpr bool ID = decltype(ident)::value;
    return p->max_row <= 6 ? (store ? assemble_step_sell_kernel<T, true, 6, ID> : assemble_step_sell_kernel<T, false, 6, ID>)
                           : (store ? assemble_step_sell_kernel<T, true, 12, ID> : assemble_step_sell_kernel<T, false, 12, ID>);
  };
  auto kern = p->rows_are_vertices ? pick(std::true_type()) : pick(std::false_type());
  hipLaunchKernelGGL(kern, dim3(p->nblocks), dim3(256), 0, st, (const T4 *)p->pos, (const uint2 *)p->topo.p, p->slice_off.p, p->idx.p,
                     p->rowg.p, (T4 *)p->coef, (const T2 *)p->state[p->cur], (T2 *)p->state[p->cur ^ 1], p->nslices, p->V, m, p->blockmax);
  hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(1024), 0, st, (const double *)p->blockmax, p->nblocks, p->maxedge);
  p->cur ^= 1;
}

// per-face records (fem_plan_face_records): not needed by the operator, computed on request
template <typename T> static hipError_t launchFaceRecords(fem_plan *p) {
  typedef typename Vec<T>::t4 T4;
  if (!p->rec) {
    hipError_t e = hipMalloc(&p->rec, 4 * sizeof(T4) * (size_t)(p->F ? p->F : 1));
    if (e != hipSuccess) return e;
  }
  double *bm = nullptr;
  hipError_t e = hipMalloc((void **)&bm, 8 * (size_t)((p->F + 255) / 256 + 1));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(assemble_faces_kernel<T>, dim3((p->F + 255) / 256), dim3(256), 0, 0, (const T4 *)p->pos, p->tri0.p,
                     p->tri1.p, p->tri2.p, (T4 *)p->rec, p->F, bm);
  e = hipDeviceSynchronize();
  (void)hipFree(bm);
  return e;
}

template <typename T> static int downloadAs(const void *dev, size_t count, double *out) {
  std::vector<T> h(count);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h.data(), dev, sizeof(T) * count, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < count; i++) out[i] = (double)h[i];
  return FEM_OK;
}

template <typename T, int C>
static hipError_t launchTile(fem_plan *p, fem_uint first, fem_uint count, const void *src, void *dst, int nsteps,
                             const ModelT<T> &m, hipStream_t stream) {
  typedef typename Vec<T>::t2 T2;
  typedef typename Vec<T>::t4 T4;
  constexpr int TPB = kTileCfgs[C].tpb, R = kTileCfgs[C].r;
  constexpr int MINW = sizeof(T) == 4 ? kTileCfgs[C].minw : (TPB == 1024 ? 4 : (TPB == 512 ? 2 : 1));
  // Both instantiations contain the wide-row loop (a build without it measured 4 % SLOWER on the
  // regular 16M sphere: same-box A/B, 204.3 vs 213.5 G DOF-it/s); they differ in where the wide
  // rows' extras are staged (kernels_step.hpp): X = 2 for plans that have any, X = 1 otherwise.
  // Read-once streams go around the caches (nontemporal) only when they cannot live in the 256 MiB
  // Infinity Cache anyway: at 1M vertices (operator 230 MB) plain loads keep the operator cached
  // between launches (247 vs 210 G DOF-it/s), from 4M on the nontemporal ones win (230 vs 223 G).
  const int pro = p->prologue >= 0 ? p->prologue : (p->tile_bytes > (256ull << 20) ? 3 : 1);   // FEM_TILE_PROLOGUE: tuning knob
  auto pick = [&](auto xtag) {
    constexpr int XV = decltype(xtag)::value;
    return pro == 3 ? tile_step_kernel<T, TPB, R, kTileW, MINW, XV, 3> : pro == 2 ? tile_step_kernel<T, TPB, R, kTileW, MINW, XV, 2>
                    : (pro == 1 ? tile_step_kernel<T, TPB, R, kTileW, MINW, XV, 1> : tile_step_kernel<T, TPB, R, kTileW, MINW, XV, 0>);
  };
  auto kern = p->maxExtra ? pick(std::integral_constant<int, 2>()) : pick(std::integral_constant<int, 1>());
  // Plans with 12-byte records exist in two prologue variants only: plain loads (1) and nontemporal loads (3);
  // FEM_TILE_PROLOGUE = 0 runs as 1 and 2 as 3 there (the knob's other values only reorder four-float loads).
  if constexpr (sizeof(T) == 4 && (C == kDefaultTileCfg || C == kDefaultTileCfgWide)) {
    if (p->packed)
      kern = p->maxExtra ? (pro >= 2 ? tile_step_kernel<T, TPB, R, kTileW, MINW, 2, 3, 1> : tile_step_kernel<T, TPB, R, kTileW, MINW, 2, 1, 1>)
                         : (pro >= 2 ? tile_step_kernel<T, TPB, R, kTileW, MINW, 1, 3, 1> : tile_step_kernel<T, TPB, R, kTileW, MINW, 1, 1, 1>);
  }
  if constexpr (sizeof(T) == 8 && (C == kDefaultTileCfg || C == kDefaultTileCfgWide)) {
    if (p->mixed)
      kern = p->maxExtra ? (pro >= 2 ? tile_step_kernel<T, TPB, R, kTileW, MINW, 2, 3, 2> : tile_step_kernel<T, TPB, R, kTileW, MINW, 2, 1, 2>)
                         : (pro >= 2 ? tile_step_kernel<T, TPB, R, kTileW, MINW, 1, 3, 2> : tile_step_kernel<T, TPB, R, kTileW, MINW, 1, 1, 2>);
  }
  const size_t lds = 2 * (size_t)p->maxLocal * sizeof(T2) + (size_t)p->maxExtra * (sizeof(T4) + 4) +
                     (size_t)TPB * R * 4 * (1 + (tileOffsetsInLds(R) ? kTileW / 2 : 0));
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3(count), dim3(TPB), lds, stream, (const T2 *)src, (T2 *)dst, p->tmeta.p,
                     p->tghost.p, p->tidx2.p, (const T4 *)p->tcoef, p->txinfo.p, first, count, p->maxLocal, p->maxExtra,
                     nsteps, m);
  return hipSuccess;
}

// `groups` groups of up to ksteps steps in ONE launch of a fixed grid of workgroups that draw (group, tile) items from a
// ticket and hand tiles over through per-tile flags (kernels_step.hpp): no chip-wide drain between groups.  Packed fp32
// plans on the two default register tiles.  A/B against one launch per group in alternating rounds of 7000 steps in one
// process (sphere, 975-row tiles, k = 7, 32 groups per launch, bit-identical states; profiles/r03_persist_rounds.txt):
// 263 tiles 176 : 176 G DOF-it/s, 1 056 tiles 388 : 370 (+4.8 %; in a cold 1000-step run as bench.py times it 310 : 286,
// the long launches get up to speed sooner), 2 387 tiles 324 : 323, 4 215 tiles 327 : 349 (-6 %), 16 848 tiles 327 : 350
// (the per-item ticket / poll / publish chain costs more than the drain of a launch that lasts 33 tile lifetimes).
// Short launches lose: 8 groups 363 : 370, 4 groups 309 : 322, 2 groups 281 : 315 at 1 056 tiles (the memset, the launch
// and its tail are paid per launch).  Hence the rule: plans of 192..2400 tiles, launches of 12..32 groups; anything
// else goes one launch per group.  FEM_TILE_PERSIST=n forces launches of 2..n groups at any size, =0 turns it off.
constexpr int kPersistGroups = 32, kPersistMinGroups = 12;
constexpr fem_uint kPersistMinTiles = 192, kPersistMaxTiles = 2400;
static int persistRule(fem_uint ntiles) { return ntiles >= kPersistMinTiles && ntiles <= kPersistMaxTiles ? kPersistGroups : 0; }
constexpr size_t kPersistCtl = 144;      // eight tickets on lines of their own, the give-up word at [128]; the flags follow
template <int C>
static hipError_t launchTilePersist(fem_plan *p, void *buf0, void *buf1, int ksteps, int lastSteps, fem_uint groups,
                                    const ModelT<float> &m, hipStream_t stream) {
  constexpr int TPB = kTileCfgs[C].tpb, R = kTileCfgs[C].r, MINW = kTileCfgs[C].minw;
  const int pro = p->prologue >= 0 ? p->prologue : (p->tile_bytes > (256ull << 20) ? 3 : 1);
  auto kern = p->maxExtra ? (pro >= 2 ? tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 2, 3, 1> : tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 2, 1, 1>)
                          : (pro >= 2 ? tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 1, 3, 1> : tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 1, 1, 1>);
  // FEM_TILE_PERSIST_COH (rate probes only; the results are wrong): 0 plain state accesses, 2 sc1 loads only, 3 sc1 stores only
  static const int cm = getenv("FEM_TILE_PERSIST_COH") ? atoi(getenv("FEM_TILE_PERSIST_COH")) : 1;
  if (!p->maxExtra && cm != 1) {
    kern = cm == 0 ? (pro >= 2 ? tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 1, 3, 1, 0> : tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 1, 1, 1, 0>)
         : cm == 2 ? (pro >= 2 ? tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 1, 3, 1, 2> : tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 1, 1, 1, 2>)
                   : (pro >= 2 ? tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 1, 3, 1, 3> : tile_step_persist_kernel<float, TPB, R, kTileW, MINW, 1, 1, 1, 3>);
  }
  size_t lds = 2 * (size_t)p->maxLocal * sizeof(float2) + (size_t)p->maxExtra * (sizeof(float4) + 4) +
               (size_t)TPB * R * 4 * (1 + (tileOffsetsInLds(R) ? kTileW / 2 : 0));
  lds = (lds + 15) & ~(size_t)15;
  const fem_uint ldsItem = (fem_uint)lds;
  lds += 16;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // control words (ticket, abort) and flags start at zero for every launch
  hipError_t e = hipMemsetAsync(p->pctl.p, 0, sizeof(fem_uint) * (kPersistCtl + (size_t)p->ntiles), stream);
  if (e != hipSuccess) return e;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // at least one workgroup for each of the eight tile ranges
  const fem_uint grid = (fem_uint)((std::max<uint64_t>(8, std::min<uint64_t>((uint64_t)p->ntiles * groups, p->persist_grid > 0 ? (uint64_t)p->persist_grid : 2ull * (uint64_t)cus)) + 7) & ~7ull);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(TPB), lds, stream, (float2 *)buf0, (float2 *)buf1, p->tmeta.p, p->tghost.p, p->tidx2.p,
                     (const float4 *)p->tcoef, p->txinfo.p, 0u, p->ntiles, p->maxLocal, p->maxExtra, ksteps, lastSteps, groups, m,
                     p->pctl.p, p->pctl.p + kPersistCtl, p->pdepOff.p, p->pdep.p, p->persist_deps, ldsItem);
  return hipSuccess;
}

// ---- semi-implicit stepping ------------------------------------------------------------------------

// comm != null: the plan is a shard's (rows [0, nOwn) owned, then ghosts): the dot products run over the owned rows and
// are summed over the ranks by comm->allreduce_sum, and every vector a matrix product reads (the right-hand side, the
// search directions) gets its ghost rows from their owners through comm->exchange first.  The iterates' ghost rows
// need no exchange of their own: x += alpha p with exchanged p and the global alpha is the owner's arithmetic.
template <typename T>
static int runImex(fem_plan *p, int nsteps, double dt, const fem_model *model, int iters, hipStream_t st,
                   const fem_imex_comm *comm = nullptr, fem_uint nOwn = 0) {
  typedef typename Vec<T>::t2 T2;
  typedef typename Vec<T>::t4 T4;
  const size_t V = p->V;
  const fem_uint Vdot = comm ? nOwn : p->V;
  const double tol2 = p->imex_tol * p->imex_tol;      // 0: fixed iteration count
  auto reduce = [&](fem_uint nparts, int what) -> int {          // partial sums -> scalars (global ones on a shard)
    if (!comm) {
      hipLaunchKernelGGL(imex_scalar_kernel, dim3(1), dim3(256), 0, st, (const double *)p->ix_partial, nparts, p->ix_sc, what, tol2);
      return FEM_OK;
    }
    hipLaunchKernelGGL(imex_reduce_kernel, dim3(1), dim3(256), 0, st, (const double *)p->ix_partial, nparts, p->ix_sc + 12,
                       (const double *)p->ix_sc, what);
    const int rc = comm->allreduce_sum(comm->ctx, (uint64_t)(p->ix_sc + 12), 2, (void *)st);
    if (rc != FEM_OK) return rc;
    hipLaunchKernelGGL(imex_finish_kernel, dim3(1), dim3(64), 0, st, (const double *)(p->ix_sc + 12), p->ix_sc, what, tol2);
    return FEM_OK;
  };
  auto ghosts = [&](void *vec) -> int { return comm ? comm->exchange(comm->ctx, (uint64_t)vec, (void *)st) : FEM_OK; };
#define IMEX_TRY(call) do { const int rc_ = (call); if (rc_ != FEM_OK) return rc_; } while (0)
  const fem_uint nbv = (fem_uint)((V + 255) / 256), nbs = p->nblocks;
  if (!p->imex_ready) {
    // work vectors: allocated into locals and committed to the plan only when every one of them exists and the
    // weights have been extracted, so that a failed allocation leaves nothing half set up for the next call
    void *bufs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sizes[6] = {sizeof(T2) * V, sizeof(T2) * V, sizeof(T2) * V, 16 * (size_t)std::max(nbv, nbs), 16 * sizeof(double),
                             sizeof(T) * (p->entries ? p->entries : 1)};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipMalloc(&bufs[i], sizes[i] ? sizes[i] : 1);
    if (e == hipSuccess) e = hipMemsetAsync(bufs[4], 0, sizes[4], st);
    if (e != hipSuccess) {
      for (void *q : bufs) if (q) (void)hipFree(q);
      return fail(FEM_ERR_DEVICE, "fem_plan_run_imex: work vectors", e);
    }
    p->ix_r = bufs[0]; p->ix_p = bufs[1]; p->ix_q = bufs[2]; p->ix_partial = (double *)bufs[3]; p->ix_sc = (double *)bufs[4]; p->ix_w = bufs[5];
    if (p->entries > 0)
      hipLaunchKernelGGL(imex_weights_kernel<T>, dim3((unsigned)((p->entries + 255) / 256)), dim3(256), 0, st, (const T4 *)p->coef,
                         (T *)p->ix_w, p->entries);
    p->imex_ready = true;
  }
  const ModelT<T> m = {(T)(model->nMax - p->base_n), (T)model->D, (T)model->r, (T)model->alpha, (T)model->S, (T)dt, (T)p->base_n,
                       (T)(1.0 / (1.0 + p->base_n))};
  const T dtDn = (T)(dt * model->D), dtDc = (T)dt;
  const T *mass = (const T *)p->mass, *wsum = (const T *)p->wsum;
  for (int s = 0; s < nsteps; s++) {
    const T2 *u = (const T2 *)p->state[p->cur];
    T2 *x = (T2 *)p->state[p->cur ^ 1];
    // b = u + dt * (explicit terms); x0 = b
    hipLaunchKernelGGL((sell_step_kernel<T, true, true>), dim3(nbs), dim3(256), 0, st, u, x, p->slice_off.p, p->idx.p,
                       (const T4 *)p->coef, p->V, p->nslices, p->nblocks, m);
    IMEX_TRY(ghosts(x));
    // r0 = M b - A b, p0 = r0 / diag, rz
    hipLaunchKernelGGL(imex_apply_kernel<T>, dim3(nbs), dim3(256), 0, st, (const T2 *)x, (T2 *)p->ix_q, (T2 *)p->ix_r, (T2 *)p->ix_p,
                       p->slice_off.p, p->idx.p, (const T *)p->ix_w, mass, wsum, p->V, Vdot, p->nslices, p->nblocks, dtDn, dtDc, 1, p->ix_partial, (const double *)p->ix_sc);
    IMEX_TRY(reduce(nbs, 0));
    IMEX_TRY(ghosts(p->ix_p));
    for (int it = 0; it < iters; it++) {
      hipLaunchKernelGGL(imex_apply_kernel<T>, dim3(nbs), dim3(256), 0, st, (const T2 *)p->ix_p, (T2 *)p->ix_q, (T2 *)nullptr, (T2 *)nullptr,
                         p->slice_off.p, p->idx.p, (const T *)p->ix_w, mass, wsum, p->V, Vdot, p->nslices, p->nblocks, dtDn, dtDc, 0, p->ix_partial, (const double *)p->ix_sc);
      IMEX_TRY(reduce(nbs, 1));
      hipLaunchKernelGGL(imex_update_kernel<T>, dim3(nbv), dim3(256), 0, st, x, (T2 *)p->ix_r, (const T2 *)p->ix_p, (const T2 *)p->ix_q,
                         mass, wsum, p->V, Vdot, dtDn, dtDc, (const double *)p->ix_sc, p->ix_partial);
      IMEX_TRY(reduce(nbv, 2));
      if (it + 1 < iters) {
        hipLaunchKernelGGL(imex_direction_kernel<T>, dim3(nbv), dim3(256), 0, st, (T2 *)p->ix_p, (const T2 *)p->ix_r, mass, wsum, p->V,
                           dtDn, dtDc, (const double *)p->ix_sc);
        IMEX_TRY(ghosts(p->ix_p));
      }
    }
    hipLaunchKernelGGL(imex_clamp_kernel<T>, dim3(nbv), dim3(256), 0, st, x, p->V, (T)p->base_n, (T)p->base_c);
    p->cur ^= 1;
  }
#undef IMEX_TRY
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

// Semi-implicit steps on a fused-tile plan (kernels_implicit.hpp, second half): one launch for the right-hand side and the
// first k - 1 Chebyshev iterations (tile_rhs_cheb_kernel: the operator rows are in registers anyway), then launches of k
// iterations each on the weights alone (tile_cheb_kernel); N iterations in all, from the spectrum bounds and the tolerance.
// comm != null: the plan is a shard's.  Every launch runs ALL local tiles (the ghost rows are iterated locally, a ring less
// valid per iteration), and the ghost rows of a vector are fetched from their owners through comm->exchange: the
// right-hand side once, the iterate and the direction after every launch but the last, the new state at the end -- 2 L
// exchanges per step for L launches, and no all-reduce: the spectrum bound `wsum` is the same number on every rank (the
// caller passes the whole mesh's), so all ranks run the same coefficient sequence.
template <typename T, int C>
static int runImexFused(fem_plan *p, int nsteps, double dt, const fem_model *model, double tol, hipStream_t st,
                        const fem_imex_comm *comm = nullptr, double wsum = 0) {
  auto ghosts = [&](void *vec) -> int { return comm ? comm->exchange(comm->ctx, (uint64_t)vec, (void *)st) : FEM_OK; };
  if (!(wsum > 0)) wsum = p->wsum_max;
  typedef typename Vec<T>::t2 T2;
  typedef typename Vec<T>::t4 T4;
  constexpr int TPB = kTileCfgs[C].tpb, R = kTileCfgs[C].r;
  constexpr int MINW = sizeof(T) == 4 ? kTileCfgs[C].minw : 2;
  const size_t V = p->V;
  if (!p->cb_tw) {
    void *bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sizes[5] = {sizeof(T) * std::max<size_t>(p->tile_entries, 1), sizeof(T2) * V, sizeof(T2) * V, sizeof(T2) * V, sizeof(T2) * V};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 5 && e == hipSuccess; i++) e = hipMalloc(&bufs[i], sizes[i]);
    if (e == hipSuccess) e = hipMemsetAsync(bufs[3], 0, sizes[3], st);
    if (e == hipSuccess) e = hipMemsetAsync(bufs[4], 0, sizes[4], st);
    if (e != hipSuccess) {
      for (void *q : bufs) if (q) (void)hipFree(q);
      return fail(FEM_ERR_DEVICE, "fem_plan_run_imex_fused: work vectors", e);
    }
    hipLaunchKernelGGL(tile_weights_kernel<T>, dim3(p->ntiles), dim3(256), 0, st, (const void *)p->tcoef, (T *)bufs[0], p->tmeta.p,
                       (fem_uint)kTileW, p->packed ? 1 : (p->mixed ? 2 : 0));
    p->cb_tw = bufs[0]; p->cb_b = bufs[1]; p->cb_x = bufs[2]; p->cb_d[0] = bufs[3]; p->cb_d[1] = bufs[4];
  }
  // iterations for the tolerance: the Chebyshev error bound 2 rho^N / (1 + rho^2N) <= tol on the wider of the two spectra
  const double Df[2] = {model->D, 1.0};
  double theta[2], delta[2];
  int N = 1;
  for (int f = 0; f < 2; f++) {
    const double hi = 1.0 + 2.0 * dt * Df[f] * wsum * 1.001;
    theta[f] = 0.5 * (hi + 1.0); delta[f] = 0.5 * (hi - 1.0);
    const double kappa = hi, rho = (std::sqrt(kappa) - 1.0) / (std::sqrt(kappa) + 1.0);
    if (rho > 0) N = std::max(N, (int)std::ceil(std::log(2.0 / tol) / -std::log(rho)));
  }
  N = std::min(N, 100000);
  p->cheb_iters = N;
  // The first launch computes the right-hand side and, from the same register rows, the first k - 1 iterations
  // (tile_rhs_cheb_kernel); FEM_IMEX_FUSED_RHS=0 (A/B knob) or k = 1: the right-hand side in a launch of its own.
  static const bool rhsKnob = !(getenv("FEM_IMEX_FUSED_RHS") && !strcmp(getenv("FEM_IMEX_FUSED_RHS"), "0"));
  const int k = p->ksteps;
  const bool fuseRhs = rhsKnob && k >= 2;
  const int nit0 = std::min(fuseRhs ? k - 1 : k, N), launches = 1 + (N - nit0 + k - 1) / k;
  const ModelT<T> m = {(T)(model->nMax - p->base_n), (T)model->D, (T)model->r, (T)model->alpha, (T)model->S, (T)dt, (T)p->base_n,
                       (T)(1.0 / (1.0 + p->base_n))};
  auto rhsKern = [&]() {
    if constexpr (sizeof(T) == 4)
      return p->packed ? (p->maxExtra ? tile_rhs_kernel<T, TPB, R, kTileW, MINW, 2, 1> : tile_rhs_kernel<T, TPB, R, kTileW, MINW, 1, 1>)
                       : (p->maxExtra ? tile_rhs_kernel<T, TPB, R, kTileW, MINW, 2, 0> : tile_rhs_kernel<T, TPB, R, kTileW, MINW, 1, 0>);
    else
      return p->mixed ? (p->maxExtra ? tile_rhs_kernel<T, TPB, R, kTileW, MINW, 2, 2> : tile_rhs_kernel<T, TPB, R, kTileW, MINW, 1, 2>)
                      : (p->maxExtra ? tile_rhs_kernel<T, TPB, R, kTileW, MINW, 2, 0> : tile_rhs_kernel<T, TPB, R, kTileW, MINW, 1, 0>);
  }();
  auto rhsChebKern = [&]() {
    if constexpr (sizeof(T) == 4)
      return p->packed ? (p->maxExtra ? tile_rhs_cheb_kernel<T, TPB, R, kTileW, MINW, 2, 1> : tile_rhs_cheb_kernel<T, TPB, R, kTileW, MINW, 1, 1>)
                       : (p->maxExtra ? tile_rhs_cheb_kernel<T, TPB, R, kTileW, MINW, 2, 0> : tile_rhs_cheb_kernel<T, TPB, R, kTileW, MINW, 1, 0>);
    else
      return p->mixed ? (p->maxExtra ? tile_rhs_cheb_kernel<T, TPB, R, kTileW, MINW, 2, 2> : tile_rhs_cheb_kernel<T, TPB, R, kTileW, MINW, 1, 2>)
                      : (p->maxExtra ? tile_rhs_cheb_kernel<T, TPB, R, kTileW, MINW, 2, 0> : tile_rhs_cheb_kernel<T, TPB, R, kTileW, MINW, 1, 0>);
  }();
  auto chebKern = p->maxExtra ? tile_cheb_kernel<T, TPB, R, kTileW, MINW, true> : tile_cheb_kernel<T, TPB, R, kTileW, MINW, false>;
  const fem_uint maxExtraEven = (p->maxExtra + 1) & ~1u;         // keeps the LDS arrays behind the iterate buffers aligned
  const size_t ldsRhs = 2 * (size_t)p->maxLocal * sizeof(T2) + (size_t)p->maxExtra * (sizeof(T4) + 4) +
                        (size_t)TPB * R * 4 * (1 + (tileOffsetsInLds(R) ? kTileW / 2 : 0));
  const size_t ldsCheb = 2 * (size_t)p->maxLocal * sizeof(T2) + (p->maxExtra ? (size_t)maxExtraEven * (sizeof(T) + 4) + (size_t)TPB * R * 4 : 0);
  if (ldsRhs > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)rhsKern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsRhs));
  if (ldsRhs > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)rhsChebKern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsRhs));
  if (ldsCheb > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)chebKern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsCheb));
  for (int s = 0; s < nsteps; s++) {
    const T2 *u = (const T2 *)p->state[p->cur];
    T2 *out = (T2 *)p->state[p->cur ^ 1];
    if (!fuseRhs) {
      hipLaunchKernelGGL(rhsKern, dim3(p->ntiles), dim3(TPB), ldsRhs, st, u, (T2 *)p->cb_b, p->tmeta.p, p->tghost.p, p->tidx2.p,
                         (const T4 *)p->tcoef, p->txinfo.p, p->ntiles, p->maxLocal, p->maxExtra, m);
      const int rc = ghosts(p->cb_b);
      if (rc != FEM_OK) return rc;
    }
    // x_0 = b; the iterate alternates between two buffers so that the last launch writes the plan's next state
    double rho[2] = {delta[0] / theta[0], delta[1] / theta[1]};
    const T2 *xin = (const T2 *)p->cb_b;
    int it = 0;
    for (int l = 0; l < launches; l++) {
      T2 *xout = ((launches - 1 - l) & 1) ? (T2 *)p->cb_x : out;
      const int nit = l == 0 ? nit0 : std::min(k, N - it);
      ChebArgs<T> a;
      a.dtDn = (T)(dt * model->D); a.dtDc = (T)dt; a.first = l == 0;
      for (int j = 0; j < 8; j++) { a.c1n[j] = a.c2n[j] = a.c1c[j] = a.c2c[j] = 0; }
      for (int j = 0; j < nit; j++, it++) {
        for (int f = 0; f < 2; f++) {
          double c1, c2;
          if (it == 0) { c1 = 0; c2 = 1.0 / theta[f]; }
          else {
            const double sigma = theta[f] / delta[f], next = 1.0 / (2.0 * sigma - rho[f]);
            c1 = next * rho[f]; c2 = 2.0 * next / delta[f];
            rho[f] = next;
          }
          (f == 0 ? a.c1n : a.c1c)[j] = (T)c1; (f == 0 ? a.c2n : a.c2c)[j] = (T)c2;
        }
      }
      if (l == 0 && fuseRhs) {
        ChebTail<T> tail;
        tail.xout = xout; tail.dout = (T2 *)p->cb_d[1]; tail.nit = nit; tail.a = a;
        hipLaunchKernelGGL(rhsChebKern, dim3(p->ntiles), dim3(TPB), ldsRhs, st, u, (T2 *)p->cb_b, p->tmeta.p, p->tghost.p, p->tidx2.p,
                           (const T4 *)p->tcoef, p->txinfo.p, p->ntiles, p->maxLocal, p->maxExtra, m, tail);
        if (launches > 1) { const int rc = ghosts(p->cb_b); if (rc != FEM_OK) return rc; }
      } else {
        hipLaunchKernelGGL(chebKern, dim3(p->ntiles), dim3(TPB), ldsCheb, st, xin, xout, (const T2 *)p->cb_b, (const T2 *)p->cb_d[l & 1],
                           (T2 *)p->cb_d[(l & 1) ^ 1], p->tmeta.p, p->tghost.p, p->tidx2.p, (const T *)p->cb_tw, p->txinfo.p, p->ntiles,
                           p->maxLocal, maxExtraEven, nit, a);
      }
      xin = xout;
      if (comm && l + 1 < launches) {
        int rc = ghosts(xout);
        if (rc == FEM_OK) rc = ghosts(p->cb_d[(l & 1) ^ 1]);
        if (rc != FEM_OK) return rc;
      }
    }
    hipLaunchKernelGGL(imex_clamp_kernel<T>, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, out, p->V, (T)p->base_n, (T)p->base_c);
    { const int rc = ghosts(out); if (rc != FEM_OK) return rc; }
    p->cur ^= 1;
  }
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

extern "C" {

#ifndef FEM_KERNEL_BUILD
#define FEM_KERNEL_BUILD "unknown"
#endif
const char *fem_kernel_build(void) { return FEM_KERNEL_BUILD; }

int fem_last_error(void) { return fem::g_last_error; }
const char *fem_error_detail(void) { return fem::g_error_detail.c_str(); }

int fem_device_synchronize(void) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

int fem_device_download(void *host_dst, uint64_t dev_src, size_t bytes) {
  if (bytes == 0) return FEM_OK;
  if (!host_dst || !dev_src) return fail(FEM_ERR_INVALID, "fem_device_download: null");
  HIP_TRY(hipMemcpy(host_dst, (const void *)dev_src, bytes, hipMemcpyDeviceToHost));
  return FEM_OK;
}

int fem_step(fem_double2 *fn_src, fem_double2 *fn_dst, fem_uint *fv, double *fv_weights,
             fem_uint fv_pitchInBytes, fem_uint fvw_pitchInBytes, fem_uint *nbr, double *vtxW,
             fem_uint vw_pitchInBytes, double *vertex_weights, fem_uint vv_pitchInBytes,
             fem_uint vws_pitchInBytes, fem_uint vv_size, fem_float4 *grads,
             fem_uint he_pitchInBytes, fem_uint *parts_n, fem_uint *block_face_count,
             fem_uint blocks, double dt, fem_uint max_part_vertices, const fem_model *model,
             void *stream) {
  if (blocks == 0) return FEM_OK;
  if (!fn_src || !fn_dst || !fv || !fv_weights || !nbr || !vtxW || !vertex_weights || !grads ||
      !parts_n || !block_face_count || fn_src == fn_dst)
    return fail(FEM_ERR_INVALID, "fem_step: null or aliased pointer");
  if (fv_pitchInBytes % sizeof(fem_uint) || fvw_pitchInBytes % sizeof(double) ||
      vw_pitchInBytes % sizeof(double) || vv_pitchInBytes % sizeof(fem_uint) ||
      vws_pitchInBytes % sizeof(double) || he_pitchInBytes % sizeof(fem_float4))
    return fail(FEM_ERR_INVALID, "fem_step: pitch not a multiple of the element size");
  // corner lists + per-face gradients in LDS (ref_step_kernel): 4 (1 + vv_size) B per vertex + 24 B per face slot
  const size_t fpitch = fv_pitchInBytes / sizeof(fem_uint);
  const size_t lds = (size_t)max_part_vertices * 4 * (1 + (size_t)vv_size) + fpitch * 24 + 16;
  if (max_part_vertices == 0 || lds > 160 * 1024)
    return fail(FEM_ERR_LIMIT, "fem_step: partition too large for the LDS work area (4 (1 + vv_size) B per vertex + 24 B per face slot <= 160 KiB)");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  ModelD m = {model->nMax, model->D, model->r, model->alpha, model->S};
  // FEM_REF_SORT=0 (measurement only): the corner lists are summed in the order the atomics filled them -- what the
  // sort costs is the difference (profiles/README.md r02: nothing measurable); results are then order-dependent
  static const bool sorted = !(getenv("FEM_REF_SORT") && !strcmp(getenv("FEM_REF_SORT"), "0"));
  auto kern = sorted ? ref_step_kernel<true> : ref_step_kernel<false>;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, (hipStream_t)stream,
                     (const double2 *)fn_src, (double2 *)fn_dst, fv, fv_weights,
                     fv_pitchInBytes / (fem_uint)sizeof(fem_uint), fvw_pitchInBytes / (fem_uint)sizeof(double),
                     nbr, vtxW, vw_pitchInBytes / (fem_uint)sizeof(double), vertex_weights,
                     vv_pitchInBytes / (fem_uint)sizeof(fem_uint), vws_pitchInBytes / (fem_uint)sizeof(double),
                     vv_size, (const float4 *)grads, he_pitchInBytes / (fem_uint)sizeof(fem_float4), parts_n,
                     block_face_count, dt, max_part_vertices, m);
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

void step(fem_double2 *fn_src, fem_double2 *fn_dst, fem_uint *fv, double *fv_weights,
          fem_uint fv_pitchInBytes, fem_uint fvw_pitchInBytes, fem_uint *nbr, double *vtxW,
          fem_uint vw_pitchInBytes, double *vertex_weights, fem_uint vv_pitchInBytes,
          fem_uint vws_pitchInBytes, fem_uint vv_size, fem_float4 *grads, fem_uint he_pitchInBytes,
          fem_uint *parts_n, fem_uint *block_face_count, fem_uint blocks, fem_uint threads,
          double dt, fem_uint smem_size) {
  (void)threads;
  // the reference caller sizes smem as 32 B per vertex of the largest partition (main.cpp:369)
  const int rc = fem_step(fn_src, fn_dst, fv, fv_weights, fv_pitchInBytes, fvw_pitchInBytes, nbr, vtxW,
                          vw_pitchInBytes, vertex_weights, vv_pitchInBytes, vws_pitchInBytes, vv_size, grads,
                          he_pitchInBytes, parts_n, block_face_count, blocks, dt, (smem_size + 31) / 32, nullptr,
                          nullptr);
  // void like the reference's: fem_last_error() right after the call is THIS call's status
  if (rc == FEM_OK) { fem::g_last_error = FEM_OK; fem::g_error_detail.clear(); }
}

// ---- reference device harness (GPUrun, main.cpp:344-452) -------------------

void fem_ref_device_destroy(fem_ref_device *d) {
  if (!d) return;
  void *ptrs[] = {d->fn_one, d->fn_two, d->vtxW, d->vv_weights, d->fv_weights, d->nbr, d->fv, d->parts_n, d->bfc, d->he_grads};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete d;
}

static hipError_t upload2D(void **dev, size_t *pitch, const void *host, size_t widthBytes, size_t rows) {
  *pitch = pitchFor(widthBytes);
  hipError_t e = hipMalloc(dev, *pitch * (rows ? rows : 1));
  if (e != hipSuccess) return e;
  e = hipMemset(*dev, 0, *pitch * (rows ? rows : 1));
  if (e != hipSuccess || rows == 0 || widthBytes == 0) return e;
  return hipMemcpy2D(*dev, *pitch, host, widthBytes, widthBytes, rows, hipMemcpyHostToDevice);
}

int fem_ref_device_create(const fem_ref_layout *lay, const fem_uint *node_parts, int V, fem_ref_device **out) {
  return fem::guarded("fem_ref_device_create", [&]() -> int {
  if (!lay || !node_parts || !out || V <= 0) return fail(FEM_ERR_INVALID, "fem_ref_device_create: bad argument");
  const fem::RefLayout &l = lay->l;
  // limits of GPUrun, main.cpp:357-364 (<= 1024 vertices and faces per partition)
  if (l.max_vertex_count > 1024 || l.max_face_count > 1024)
    return fail(FEM_ERR_LIMIT, "ERROR: Too many nodes or elements per Block! (> 1024)");
  fem_ref_device *d = new fem_ref_device();
  d->n = l.n; d->V = V; d->maxV = l.max_vertex_count; d->W = l.vv_max_neighbors; d->maxF = l.max_face_count;
  d->threads = std::max((d->maxV + 31) / 32 * 32, (d->maxF + 31) / 32 * 32);   // main.cpp:366-368
  d->smem = d->maxV * 32;                                                        // main.cpp:369
  const size_t n = (size_t)l.n;
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
  ok(hipMalloc((void **)&d->fn_one, sizeof(double2) * (size_t)V)) &&
  ok(hipMalloc((void **)&d->fn_two, sizeof(double2) * (size_t)V)) &&
  ok(hipMalloc((void **)&d->parts_n, sizeof(fem_uint) * (n + 1))) &&
  ok(hipMalloc((void **)&d->bfc, sizeof(fem_uint) * n)) &&
  ok(hipMemcpy(d->parts_n, node_parts, sizeof(fem_uint) * (n + 1), hipMemcpyHostToDevice)) &&
  ok(hipMemcpy(d->bfc, l.block_face_count.data(), sizeof(fem_uint) * n, hipMemcpyHostToDevice)) &&
  ok(upload2D((void **)&d->vtxW, &d->vw_pitch, l.vertex_weights.data(), d->maxV * sizeof(double), n)) &&
  ok(upload2D((void **)&d->nbr, &d->nbr_pitch, l.nbr_v.data(), d->maxV * sizeof(fem_uint), n * (d->W + 1))) &&
  ok(upload2D((void **)&d->vv_weights, &d->vvw_pitch, l.vv_weights.data(), d->maxV * sizeof(double), n * d->W)) &&
  ok(upload2D((void **)&d->he_grads, &d->he_pitch, l.he_grads.data(), d->maxF * sizeof(float4), n * 2)) &&
  ok(upload2D((void **)&d->fv, &d->fv_pitch, l.face_vertices.data(), d->maxF * sizeof(fem_uint), n * 3)) &&
  ok(upload2D((void **)&d->fv_weights, &d->fvw_pitch, l.fv_weights_new.data(), d->maxF * sizeof(double), n * 3));
  if (e != hipSuccess) { fem_ref_device_destroy(d); return fail(FEM_ERR_DEVICE, "fem_ref_device_create", e); }
  *out = d;
  return FEM_OK;
  });
}

int fem_ref_device_set_state(fem_ref_device *d, const fem_double2 *fn) {
  if (!d || !fn) return fail(FEM_ERR_INVALID, "fem_ref_device_set_state: null");
  HIP_TRY(hipMemcpy(d->fn_one, fn, sizeof(double2) * (size_t)d->V, hipMemcpyHostToDevice));
  d->one_is_current = true;
  return FEM_OK;
}

int fem_ref_device_get_state(fem_ref_device *d, fem_double2 *fn) {
  if (!d || !fn) return fail(FEM_ERR_INVALID, "fem_ref_device_get_state: null");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(fn, d->one_is_current ? d->fn_one : d->fn_two, sizeof(double2) * (size_t)d->V, hipMemcpyDeviceToHost));
  return FEM_OK;
}

int fem_ref_device_run(fem_ref_device *d, int nsteps, double dt, const fem_model *model, void *stream) {
  if (!d || nsteps < 0) return fail(FEM_ERR_INVALID, "fem_ref_device_run: bad argument");
  for (int i = 0; i < nsteps; i++) {
    double2 *src = d->one_is_current ? d->fn_one : d->fn_two, *dst = d->one_is_current ? d->fn_two : d->fn_one;
    int rc = fem_step((fem_double2 *)src, (fem_double2 *)dst, d->fv, d->fv_weights, (fem_uint)d->fv_pitch,
                      (fem_uint)d->fvw_pitch, d->nbr, d->vtxW, (fem_uint)d->vw_pitch, d->vv_weights,
                      (fem_uint)d->nbr_pitch, (fem_uint)d->vvw_pitch, d->W, (fem_float4 *)d->he_grads,
                      (fem_uint)d->he_pitch, d->parts_n, d->bfc, (fem_uint)d->n, dt, d->maxV, model, stream);
    if (rc != FEM_OK) return rc;
    d->one_is_current = !d->one_is_current;
  }
  return FEM_OK;
}

void fem_ref_device_args(const fem_ref_device *d, uint64_t *fn_one, uint64_t *fn_two, uint64_t *fv,
                         uint64_t *fv_weights, fem_uint *fv_pitch, fem_uint *fvw_pitch, uint64_t *nbr,
                         uint64_t *vtxW, fem_uint *vw_pitch, uint64_t *vertex_weights, fem_uint *vv_pitch,
                         fem_uint *vws_pitch, fem_uint *vv_size, uint64_t *grads, fem_uint *he_pitch,
                         uint64_t *parts_n, uint64_t *block_face_count, fem_uint *blocks,
                         fem_uint *threads, fem_uint *smem_size) {
  *fn_one = (uint64_t)d->fn_one; *fn_two = (uint64_t)d->fn_two;
  *fv = (uint64_t)d->fv; *fv_weights = (uint64_t)d->fv_weights;
  *fv_pitch = (fem_uint)d->fv_pitch; *fvw_pitch = (fem_uint)d->fvw_pitch;
  *nbr = (uint64_t)d->nbr; *vtxW = (uint64_t)d->vtxW; *vw_pitch = (fem_uint)d->vw_pitch;
  *vertex_weights = (uint64_t)d->vv_weights; *vv_pitch = (fem_uint)d->nbr_pitch;
  *vws_pitch = (fem_uint)d->vvw_pitch; *vv_size = d->W;
  *grads = (uint64_t)d->he_grads; *he_pitch = (fem_uint)d->he_pitch;
  *parts_n = (uint64_t)d->parts_n; *block_face_count = (uint64_t)d->bfc;
  *blocks = (fem_uint)d->n; *threads = d->threads; *smem_size = d->smem;
}

// ---- native plan ------------------------------------------------------------

void fem_plan_destroy(fem_plan *p) {
  if (!p) return;
  p->slice_off.release(); p->idx.release(); p->send_idx.release();
  p->tmeta.release(); p->tghost.release(); p->tidx2.release(); p->txinfo.release();
  p->pctl.release(); p->pdepOff.release(); p->pdep.release();
  p->tri0.release(); p->tri1.release(); p->tri2.release(); p->topo.release(); p->rowg.release();
  for (void *q : {p->pos, p->pos0, p->rec, (void *)p->maxedge, (void *)p->blockmax}) if (q) (void)hipFree(q);
  for (void *q : {p->mass, p->wsum, p->ix_w, p->ix_r, p->ix_p, p->ix_q, (void *)p->ix_partial, (void *)p->ix_sc}) if (q) (void)hipFree(q);
  for (void *q : {p->cb_tw, p->cb_b, p->cb_x, p->cb_d[0], p->cb_d[1]}) if (q) (void)hipFree(q);
  if (p->tcoef) (void)hipFree(p->tcoef);
  if (p->rscale) (void)hipFree(p->rscale);
  if (p->coef) (void)hipFree(p->coef);
  for (void *s : p->state) if (s) (void)hipFree(s);
  delete p;
}

static int planCreate(const fem_mesh *ordered, const fem::Shard *shard, int precision, const fem_model *model,
                      int ksteps, int n, const fem_uint *node_parts, unsigned flags, fem_plan **out) {
  if (!ordered || !out || (precision != FEM_PREC_F32 && precision != FEM_PREC_F64) || ksteps < 0 || ksteps > 8)
    return fail(FEM_ERR_INVALID, "fem_plan_create: bad argument");
  if (ksteps > 0 && (!node_parts || n < 1)) return fail(FEM_ERR_INVALID, "fem_plan_create_tiled: partition ranges missing");
  if ((flags & FEM_PLAN_IMPLICIT) && ksteps > 0)
    return fail(FEM_ERR_INVALID, "fem_plan_create_ex: FEM_PLAN_IMPLICIT needs a sliced-ELL plan (ksteps = 0)");
  if ((flags & FEM_PLAN_IMPLICIT) && (flags & FEM_PLAN_DEVICE_ASSEMBLY))
    return fail(FEM_ERR_INVALID, "fem_plan_create_ex: FEM_PLAN_IMPLICIT takes the lumped masses from the host assembly; not combinable with FEM_PLAN_DEVICE_ASSEMBLY");
  if ((flags & FEM_PLAN_MIXED_RECORDS) && (precision != FEM_PREC_F64 || ksteps < 1 || (flags & (FEM_PLAN_DEVICE_ASSEMBLY | FEM_PLAN_IMPLICIT))))
    return fail(FEM_ERR_INVALID, "fem_plan_create_ex: FEM_PLAN_MIXED_RECORDS needs a fused fp64 plan (ksteps >= 1) whose operator comes from the host assembly");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  if (!(model->nMax > -1.0)) return fail(FEM_ERR_INVALID, "fem_plan_create: nMax must be > -1");
  fem::RowOp op;
  // plans that assemble on the device keep their rows in fan order (assemble_row_fan); the others
  // keep the ascending order their measured kernels were tuned on
  const bool fan = (flags & FEM_PLAN_DEVICE_ASSEMBLY) != 0;
  int rc = shard ? fem::buildRowOp(ordered->mesh, ordered->stats, op, shard->loc2glob.data(), (fem::uint)shard->loc2glob.size(), fan)
                 : fem::buildRowOp(ordered->mesh, ordered->stats, op, nullptr, 0, fan);
  if (rc != FEM_OK) return fail(rc, "fem_plan_create: mesh has no topology/stats or is too large");
  fem::SellLayout L;
  fem::TileLayout TL;
  std::vector<unsigned char> tileClass;
  int tcfg = 0;
  if (ksteps > 0) {
    // tiles = this rank's partitions, then (shards only) the ghost rows in pieces
    std::vector<std::pair<fem::uint, fem::uint>> ranges;
    const fem::uint own_b = shard ? shard->own_begin : 0, own_e = shard ? shard->own_end : op.V;
    for (int p = 0; p < n; p++)
      if (node_parts[p] >= own_b && node_parts[p + 1] <= own_e && node_parts[p + 1] > node_parts[p])
        ranges.push_back(std::make_pair(node_parts[p] - own_b, node_parts[p + 1] - own_b));
    fem::uint covered = 0;
    for (auto &r : ranges) covered += r.second - r.first;
    if (covered != own_e - own_b) return fail(FEM_ERR_INVALID, "fem_plan_create_tiled: partitions do not tile the owned range");
    for (fem::uint g = own_e - own_b; g < op.V; g += 512) ranges.push_back(std::make_pair(g, std::min(op.V, g + 512)));
    const char *cfgEnv = getenv("FEM_TILE_CONFIG");
    // default geometry: 512 threads x 4 register rows (2048-row tiles) when no row is wider than the
    // register tile's 6 entries; meshes with wide rows also stage extras in LDS, where two workgroups
    // per CU only fit with the 3-row tile
    bool wideRows = false;
    for (fem::uint r = 0; r < op.V && !wideRows; r++) wideRows = op.rowptr[r + 1] - op.rowptr[r] > kTileW;
    // (small meshes: the 3-row tile too -- more, smaller tiles fill the 512 workgroup slots better -- unless the caller
    //  has cut partitions for the 4-row tile, > 800 rows on average, and fem_plan_run will hand the tiles over inside
    //  persistent launches, which fill the slots whatever the tile count: sphere1m 313 G against 300 G with 3-row tiles)
    const fem::uint nOwnedParts = (fem::uint)ranges.size() - (fem::uint)((op.V - (own_e - own_b) + 511) / 512);
    const bool bigParts = nOwnedParts > 0 && (own_e - own_b) / nOwnedParts > 800;
    const char *perEnv = getenv("FEM_TILE_PERSIST"), *pk0 = getenv("FEM_TILE_PACKED");
    const bool persistLikely = !shard && precision == FEM_PREC_F32 && !(flags & FEM_PLAN_DEVICE_ASSEMBLY) && !(pk0 && !strcmp(pk0, "0")) &&
                               (perEnv ? atoi(perEnv) > 0 : persistRule((own_e - own_b + 974) / 975) > 0);
    tcfg = cfgEnv ? atoi(cfgEnv)
                  : (wideRows || (op.V < kLargeMeshRows && !(bigParts && persistLikely)) ? kDefaultTileCfgWide : kDefaultTileCfg);
    if (tcfg < 0 || tcfg >= kNumTileCfgs) return fail(FEM_ERR_INVALID, "FEM_TILE_CONFIG out of range");
    rc = fem::packTileLayout(op, ranges, ksteps, kTileCfgs[tcfg].tpb, kTileCfgs[tcfg].r, kTileW,
                             precision == FEM_PREC_F32 ? kTileMaxLocal : kTileMaxLocal * 3 / 4,
                             precision == FEM_PREC_F32 ? kTileMaxExtra : kTileMaxExtra * 2 / 3, TL);   // LDS <= 160 KiB
    if (rc == FEM_ERR_LIMIT) return fail(rc, "fem_plan_create_tiled: a partition cannot be cut into tiles that fit the fused kernel");
    if (rc != FEM_OK) return fail(rc, "fem_plan_create_tiled: cannot tile");
    // Shards: order the tiles [interior | boundary | ghost] so that a launch can take a class
    // range (DESIGN.md "Multi-GPU": exchange overlapped with the interior tiles).
    //   ghost    = tiles of ghost rows (their last group before an exchange is never consumed);
    //   boundary = owned tiles that stage a ghost row (need the exchange before they run) or hold
    //              a row another rank needs (must be done before the exchange is packed);
    //   interior = the rest.
    tileClass.assign(TL.ntiles, 0);
    if (shard) {
      const fem::uint nOwn = own_e - own_b;
      std::vector<unsigned char> isSend(op.V, 0);
      for (fem::uint i : shard->send_idx) if (i < op.V) isSend[i] = 1;
      for (fem::uint t = 0; t < TL.ntiles; t++) {
        const fem::uint *mt = &TL.meta[16 * (size_t)t];
        const fem::uint base = mt[0], goff = mt[1], k = mt[4], m0 = mt[5], mk = mt[5 + k];
        if (base >= nOwn) { tileClass[t] = 2; continue; }
        bool bnd = false;
        for (fem::uint i = 0; i < mk - m0 && !bnd; i++) bnd = TL.ghost[goff + i] >= nOwn;
        for (fem::uint i = 0; i < m0 && !bnd; i++) bnd = isSend[base + i] != 0;
        tileClass[t] = bnd ? 1 : 0;
      }
      std::vector<fem::uint> order(TL.ntiles), meta2(TL.meta.size());
      for (fem::uint t = 0; t < TL.ntiles; t++) order[t] = t;
      std::stable_sort(order.begin(), order.end(), [&](fem::uint a, fem::uint b) { return tileClass[a] < tileClass[b]; });
      for (fem::uint t = 0; t < TL.ntiles; t++)
        std::copy(&TL.meta[16 * (size_t)order[t]], &TL.meta[16 * (size_t)order[t]] + 16, &meta2[16 * (size_t)t]);
      TL.meta.swap(meta2);
    }
  } else {
    rc = fem::packSellLayout(op, L);
    if (rc != FEM_OK) return fail(rc, "fem_plan_create: mesh too large");
  }
  fem_plan *p = new fem_plan();
  (void)hipGetDevice(&p->device);
  p->base_n = model->nMax; p->base_c = model->nMax / (1 + model->nMax);
  p->precision = precision; p->V = op.V;
  // half a rigorous bound on the spectrum of -L (the Chebyshev solver's interval): Gershgorin on the rows as packed,
  // or the per-triangle bound of the mesh's stats where that is smaller (a whole-mesh plan; shards are given theirs)
  for (fem::uint r = 0; r < op.V; r++) {
    double ws = 0;
    for (fem::uint q = op.rowptr[r]; q < op.rowptr[r + 1]; q++) ws += std::fabs(op.coef[4 * (size_t)q]);
    if (ws > p->wsum_max) p->wsum_max = ws;
  }
  if (!shard) {
    const double sb = fem::laplacianSpectrumBound(ordered->mesh, ordered->stats);
    if (sb > 0 && 0.5 * sb < p->wsum_max) p->wsum_max = 0.5 * sb;
  }
  if (const char *e = getenv("FEM_TILE_PROLOGUE")) p->prologue = std::min(3, std::max(0, atoi(e)));
  if (const char *e = getenv("FEM_SELL_NT")) p->sell_nt = atoi(e) != 0;
  if (const char *e = getenv("FEM_TILE_PERSIST")) p->persist = std::max(0, atoi(e));      // 0 off, n: n groups per launch, whatever the size
  if (const char *e = getenv("FEM_TILE_PERSIST_DEPS")) p->persist_deps = atoi(e) != 0;
  if (const char *e = getenv("FEM_TILE_PERSIST_GRID")) p->persist_grid = std::max(0, atoi(e));
  const size_t sb = precision == FEM_PREC_F32 ? sizeof(float) : sizeof(double);
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
  auto uploadCoef = [&](void **dev, const std::vector<double> &c) {
    if (!ok(hipMalloc(dev, sb * (c.empty() ? 4 : c.size())))) return;
    if (precision == FEM_PREC_F32) {
      std::vector<float> c32(c.begin(), c.end());
      ok(hipMemcpy(*dev, c32.data(), sizeof(float) * c32.size(), hipMemcpyHostToDevice));
    } else {
      ok(hipMemcpy(*dev, c.data(), sizeof(double) * c.size(), hipMemcpyHostToDevice));
    }
  };
  if (ksteps > 0) {
    p->ksteps = ksteps; p->ntiles = TL.ntiles; p->tcfg = tcfg; p->tile_rows = TL.rowSteps;
    for (unsigned char c : tileClass) (c == 0 ? p->nInterior : (c == 1 ? p->nBoundary : p->nGhost))++;
    p->maxLocal = (TL.maxLocal + 1) & ~1u; p->maxExtra = TL.maxExtra; p->tile_entries = TL.coef.size() / 4;
    p->tile_bytes = (uint64_t)(TL.coef.size() / 4) * (4 * sb + 2) + (uint64_t)TL.ghost.size() * 4 + (uint64_t)TL.meta.size() * 4 +
                    (uint64_t)TL.xinfo.size() * 4;
    if (p->persist < 0) { p->persist = persistRule(TL.ntiles); p->persist_min = kPersistMinGroups; }
    if (p->persist > 0) {
      // tiles each tile waits for: the owners of its ring rows, and itself (another workgroup may run its next group)
      std::vector<fem_uint> owner(op.V, 0u), depOff(TL.ntiles + 1, 0u), dep;
      for (fem_uint t = 0; t < TL.ntiles; t++) {
        const fem_uint *mt = &TL.meta[16 * (size_t)t];
        for (fem_uint i = 0; i < mt[5]; i++) owner[mt[0] + i] = t;
      }
      // (made symmetric: a tile whose rings are cut off at a shard's edge may not reach a tile that reaches it)
      std::vector<std::vector<fem_uint>> nb(TL.ntiles);
      for (fem_uint t = 0; t < TL.ntiles; t++) {
        const fem_uint *mt = &TL.meta[16 * (size_t)t];
        const fem_uint cnt = mt[5 + mt[4]] - mt[5];
        nb[t].push_back(t);
        for (fem_uint i = 0; i < cnt; i++) {
          const fem_uint o = owner[TL.ghost[mt[1] + i]];
          if (nb[t].back() != o) { nb[t].push_back(o); nb[o].push_back(t); }
        }
      }
      for (fem_uint t = 0; t < TL.ntiles; t++) {
        std::vector<fem_uint> &tmp = nb[t];
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        dep.insert(dep.end(), tmp.begin(), tmp.end());
        depOff[t + 1] = (fem_uint)dep.size();
      }
      ok(p->pctl.alloc(kPersistCtl + (size_t)TL.ntiles)) && ok(hipMemset(p->pctl.p, 0, sizeof(fem_uint) * (kPersistCtl + (size_t)TL.ntiles))) &&
      ok(p->pdepOff.alloc(depOff.size())) && ok(p->pdepOff.upload(depOff.data())) &&
      ok(p->pdep.alloc(dep.size())) && ok(p->pdep.upload(dep.data()));
    }
    ok(p->tmeta.alloc(TL.meta.size())) && ok(p->tmeta.upload(TL.meta.data())) &&
    ok(p->tghost.alloc(TL.ghost.size())) && (TL.ghost.empty() || ok(p->tghost.upload(TL.ghost.data()))) &&
    ok(p->tidx2.alloc(TL.idx2.size())) && ok(p->tidx2.upload(TL.idx2.data())) &&
    ok(p->txinfo.alloc(TL.xinfo.size())) && ok(p->txinfo.upload(TL.xinfo.data()));
    // 12-byte operator records (FEM_TILE_PACKED, default on): fp32 plans of the two default tile geometries whose
    // operator comes from the host assembly.  The array keeps the four-float layout's size and tile offsets
    // (kernels_step.hpp, PK): per tile W * rows_pad records, then rows_pad scales; extras stay four floats.
    const char *pkEnv = getenv("FEM_TILE_PACKED");
    bool packed = precision == FEM_PREC_F32 && !(flags & FEM_PLAN_DEVICE_ASSEMBLY) &&
                  (tcfg == kDefaultTileCfg || tcfg == kDefaultTileCfgWide) && !(pkEnv && !strcmp(pkEnv, "0"));
    if (packed) {
      std::vector<float> c32(TL.coef.begin(), TL.coef.end());
      fem::uint *words = reinterpret_cast<fem::uint *>(c32.data());
      constexpr int kQ = (1 << 20) - 1;
      std::vector<unsigned char> badTile(TL.ntiles, 0);      // a coefficient that is not finite (degenerate triangles): keep four floats, NaN and all
      uint64_t recWords = 0, scaleWords = 0;
      for (fem::uint t = 0; t < TL.ntiles; t++) { recWords += (uint64_t)kTileW * TL.meta[16 * (size_t)t + 3]; scaleWords += TL.meta[16 * (size_t)t + 3]; }
      fem::parallelFor(TL.ntiles, 64, [&](size_t t0, size_t t1) {
        std::vector<fem::uint> block;
        for (size_t t = t0; t < t1; t++) {
          const size_t roff = TL.meta[16 * t + 2], rpad = TL.meta[16 * t + 3];
          block.assign(4 * (size_t)kTileW * rpad, 0u);
          for (size_t r = 0; r < rpad; r++) {
            double big = 0;
            for (size_t w = 0; w < (size_t)kTileW; w++)
              for (int d = 1; d < 4; d++) big = std::max(big, std::fabs(TL.coef[4 * (roff + w * rpad + r) + d]));
            if (!std::isfinite(big)) badTile[t] = 1;
            const float sc = std::isfinite(big) ? (float)(big / kQ) : 0.0f;
            memcpy(&block[3 * (size_t)kTileW * rpad + r], &sc, 4);
            for (size_t w = 0; w < (size_t)kTileW; w++) {
              const size_t e = roff + w * rpad + r, le = w * rpad + r;
              const float wt = (float)TL.coef[4 * e];
              fem::uint q[3] = {0, 0, 0};
              for (int d = 0; d < 3; d++) {
                long v = sc > 0 ? std::lrint(TL.coef[4 * e + 1 + d] / (double)sc) : 0;
                v = std::max<long>(-kQ, std::min<long>(kQ, v));
                q[d] = (fem::uint)v & 0x1fffffu;
              }
              memcpy(&block[3 * le], &wt, 4);
              block[3 * le + 1] = q[0] | (q[1] << 21);
              block[3 * le + 2] = (q[1] >> 11) | (q[2] << 10);
            }
          }
          memcpy(words + 4 * roff, block.data(), 4 * block.size());
        }
      });
      for (unsigned char b : badTile) if (b) packed = false;
      if (packed) {
        ok(hipMalloc(&p->tcoef, 4 * std::max<size_t>(c32.size(), 4))) &&
        ok(hipMemcpy(p->tcoef, c32.data(), 4 * c32.size(), hipMemcpyHostToDevice));
        p->packed = true;
        p->tile_bytes -= recWords * 4;                       // 12-byte records ...
        p->tile_bytes += scaleWords * 4;                     // ... and one scale per row
      }
    }
    if (flags & FEM_PLAN_MIXED_RECORDS) {
      // The reference's own mix (src/common.h:7-11, FEM_common.cu:82-166: double2 state and Laplacian, float4 gradient
      // vectors, float3 accumulators) on the fused tiles: per tile W * rows_pad weights in fp64, then W * rows_pad gradient
      // coefficients of three floats, inside the block the four-double layout gives the tile (20 of its 32 bytes per
      // entry are read); the extras of wide rows stay four doubles behind the block.
      if (tcfg != kDefaultTileCfg && tcfg != kDefaultTileCfgWide) {
        fem_plan_destroy(p);
        return fail(FEM_ERR_INVALID, "fem_plan_create_ex: FEM_PLAN_MIXED_RECORDS runs on the two default register tiles (FEM_TILE_CONFIG 2 or 8)");
      }
      std::vector<double> mixedCoef(TL.coef);
      uint64_t blockEntries = 0;
      for (fem::uint t = 0; t < TL.ntiles; t++) blockEntries += (uint64_t)kTileW * TL.meta[16 * (size_t)t + 3];
      fem::parallelFor(TL.ntiles, 64, [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; t++) {
          const size_t roff = TL.meta[16 * t + 2], rpad = TL.meta[16 * t + 3], n = (size_t)kTileW * rpad;
          double *block = mixedCoef.data() + 4 * roff;
          std::fill(block, block + 4 * n, 0.0);
          unsigned char *g = reinterpret_cast<unsigned char *>(block + n);      // the float triples, written bytewise
          for (size_t le = 0; le < n; le++) {
            block[le] = TL.coef[4 * (roff + le)];
            const float c3[3] = {(float)TL.coef[4 * (roff + le) + 1], (float)TL.coef[4 * (roff + le) + 2], (float)TL.coef[4 * (roff + le) + 3]};
            memcpy(g + 12 * le, c3, 12);
          }
        }
      });
      uploadCoef(&p->tcoef, mixedCoef);
      p->mixed = true;
      p->tile_bytes -= blockEntries * 12;
    } else if (!packed) uploadCoef(&p->tcoef, TL.coef);
  } else {
    p->nslices = L.nslices; p->nblocks = (L.nslices + 3) / 4; p->entries = L.idx.size();
    ok(p->slice_off.alloc(L.slice_off.size())) && ok(p->slice_off.upload(L.slice_off.data())) &&
    ok(p->idx.alloc(L.idx.size())) && ok(p->idx.upload(L.idx.data()));
    // 12-byte operator records (FEM_TILE_PACKED, default on) for fp32 plans whose operator comes from the host
    // assembly and that do no implicit stepping (whose kernels read the weights out of four-float entries)
    const char *pkEnv = getenv("FEM_TILE_PACKED");
    bool packed = precision == FEM_PREC_F32 && flags == 0 && !L.idx.empty() && !(pkEnv && !strcmp(pkEnv, "0"));
    if (packed) {
      const size_t entries = L.idx.size();
      std::vector<fem::uint> rec(3 * entries, 0u);
      std::vector<float> scale(op.V, 0.0f);
      constexpr int kQ = (1 << 20) - 1;
      std::vector<unsigned char> bad(L.nslices, 0);
      fem::parallelFor(L.nslices, 256, [&](size_t s0, size_t s1) {
        for (size_t sl = s0; sl < s1; sl++) {
          const size_t off = L.slice_off[sl], width = (L.slice_off[sl + 1] - off) / 64;
          for (size_t lane = 0; lane < 64; lane++) {
            double big = 0;
            for (size_t k = 0; k < width; k++)
              for (int d = 1; d < 4; d++) big = std::max(big, std::fabs(L.coef[4 * (off + 64 * k + lane) + d]));
            if (!std::isfinite(big)) bad[sl] = 1;
            const float sc = std::isfinite(big) ? (float)(big / kQ) : 0.0f;
            const size_t v = 64 * sl + lane;
            if (v < op.V) scale[v] = sc;
            for (size_t k = 0; k < width; k++) {
              const size_t e = off + 64 * k + lane;
              const float wt = (float)L.coef[4 * e];
              fem::uint q[3] = {0, 0, 0};
              for (int d = 0; d < 3; d++) {
                long x = sc > 0 ? std::lrint(L.coef[4 * e + 1 + d] / (double)sc) : 0;
                x = std::max<long>(-kQ, std::min<long>(kQ, x));
                q[d] = (fem::uint)x & 0x1fffffu;
              }
              memcpy(&rec[3 * e], &wt, 4);
              rec[3 * e + 1] = q[0] | (q[1] << 21);
              rec[3 * e + 2] = (q[1] >> 11) | (q[2] << 10);
            }
          }
        }
      });
      for (unsigned char b : bad) if (b) packed = false;      // a coefficient that is not finite: keep four floats, NaN and all
      if (packed) {
        ok(hipMalloc(&p->coef, 4 * rec.size())) && ok(hipMemcpy(p->coef, rec.data(), 4 * rec.size(), hipMemcpyHostToDevice)) &&
        ok(hipMalloc((void **)&p->rscale, 4 * scale.size())) && ok(hipMemcpy(p->rscale, scale.data(), 4 * scale.size(), hipMemcpyHostToDevice));
        p->packed = true;
      }
    }
    if (!packed) uploadCoef(&p->coef, L.coef);
  }
  ok(hipMalloc(&p->state[0], 2 * sb * p->V)) && ok(hipMalloc(&p->state[1], 2 * sb * p->V));
  if (e == hipSuccess && (flags & FEM_PLAN_DEVICE_ASSEMBLY)) {
    const fem::SimpleTriMesh &m = ordered->mesh;
    p->F = m.n_faces(); p->Vg = m.n_vertices(); p->can_assemble = true;
    p->fan_rows = op.fanOrdered && getenv("FEM_ASSEMBLE_GENERIC") == nullptr;
    for (fem::uint r = 0; r < op.V; r++) p->max_row = std::max<fem_uint>(p->max_row, op.rowptr[r + 1] - op.rowptr[r]);
    p->wide_rows = p->max_row > 6;
    p->rows_are_vertices = op.V == p->Vg;
    for (fem::uint r = 0; r < op.V && p->rows_are_vertices; r++) p->rows_are_vertices = op.rowg[r] == r;
    std::vector<fem_uint> t0(p->F), t1(p->F), t2(p->F);
    for (fem_uint f = 0; f < p->F; f++) { t0[f] = m.faces[3 * (size_t)f]; t1[f] = m.faces[3 * (size_t)f + 1]; t2[f] = m.faces[3 * (size_t)f + 2]; }
    const std::vector<fem_uint> &topo = ksteps > 0 ? TL.topo : L.topo;
    ok(p->tri0.alloc(p->F)) && ok(p->tri0.upload(t0.data())) && ok(p->tri1.alloc(p->F)) && ok(p->tri1.upload(t1.data())) &&
    ok(p->tri2.alloc(p->F)) && ok(p->tri2.upload(t2.data())) && ok(p->topo.alloc(topo.size())) && ok(p->topo.upload(topo.data())) &&
    ok(p->rowg.alloc(op.rowg.size())) && ok(p->rowg.upload(op.rowg.data())) &&
    ok(hipMalloc(&p->pos, 4 * sb * p->Vg)) && ok(hipMalloc(&p->pos0, 4 * sb * p->Vg)) &&
    ok(hipMalloc((void **)&p->maxedge, 8)) && ok(hipMemset(p->maxedge, 0, 8)) &&
    ok(hipMalloc((void **)&p->blockmax, 2 * 8 * (size_t)(std::max<size_t>(ksteps > 0 ? TL.ntiles : (L.nslices + 3) / 4, 1))));
  }
  if (e == hipSuccess && (flags & FEM_PLAN_IMPLICIT)) {
    // lumped mass a_v and sum_j w_vj of every row (MeshStats: areaPerVtx, -wVtx)
    std::vector<double> hm(op.V), hw(op.V);
    for (fem::uint r = 0; r < op.V; r++) { hm[r] = ordered->stats.areaPerVtx[op.rowg[r]]; hw[r] = -ordered->stats.wVtx[op.rowg[r]]; }
    auto uploadVec = [&](void **dev, const std::vector<double> &h) {
      if (!ok(hipMalloc(dev, sb * h.size()))) return;
      if (precision == FEM_PREC_F32) { std::vector<float> f(h.begin(), h.end()); ok(hipMemcpy(*dev, f.data(), 4 * f.size(), hipMemcpyHostToDevice)); }
      else ok(hipMemcpy(*dev, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
    };
    uploadVec(&p->mass, hm); uploadVec(&p->wsum, hw);
    p->can_imex = e == hipSuccess;
  }
  if (e == hipSuccess && shard)
    ok(p->send_idx.alloc(shard->send_idx.size())) && (shard->send_idx.empty() || ok(p->send_idx.upload(shard->send_idx.data())));
  if (e == hipSuccess) {
    ok(hipMemset(p->state[0], 0, 2 * sb * p->V));
    ok(hipMemset(p->state[1], 0, 2 * sb * p->V));
  }
  if (e != hipSuccess) { fem_plan_destroy(p); return fail(FEM_ERR_DEVICE, "fem_plan_create", e); }
  *out = p;
  return FEM_OK;
}

int fem_plan_create(const fem_mesh *ordered, int precision, const fem_model *model, fem_plan **out) {
  return fem::guarded("fem_plan_create", [&]() -> int {
  return planCreate(ordered, nullptr, precision, model, 0, 0, nullptr, 0, out);
  });
}

int fem_plan_create_ex(const fem_mesh *ordered, int n, const fem_uint *node_parts, const fem_shard *shard,
                       int precision, const fem_model *model, int ksteps, unsigned flags, fem_plan **out) {
  return fem::guarded("fem_plan_create_ex", [&]() -> int {
  int rc = planCreate(ordered, shard ? &shard->s : nullptr, precision, model, ksteps, n, node_parts, flags, out);
  if (rc == FEM_OK && (flags & FEM_PLAN_DEVICE_ASSEMBLY)) {
    rc = fem_plan_set_points(*out, ordered->mesh.points.data());
    if (rc != FEM_OK) { fem_plan_destroy(*out); *out = nullptr; }
  }
  return rc;
  });
}

int fem_plan_create_tiled(const fem_mesh *ordered, int n, const fem_uint *node_parts, const fem_shard *shard,
                          int precision, const fem_model *model, int ksteps, fem_plan **out) {
  return fem::guarded("fem_plan_create_tiled", [&]() -> int {
  if (ksteps < 1) return fail(FEM_ERR_INVALID, "fem_plan_create_tiled: ksteps must be 1..8");
  return planCreate(ordered, shard ? &shard->s : nullptr, precision, model, ksteps, n, node_parts, 0, out);
  });
}

int fem_plan_create_shard(const fem_mesh *ordered, const fem_shard *shard, int precision, const fem_model *model,
                          fem_plan **out) {
  return fem::guarded("fem_plan_create_shard", [&]() -> int {
  if (!shard) return fail(FEM_ERR_INVALID, "fem_plan_create_shard: null shard");
  return planCreate(ordered, &shard->s, precision, model, 0, 0, nullptr, 0, out);
  });
}

int fem_plan_pack_halo(fem_plan *p, uint64_t dst_dev, void *stream) { return fem_plan_pack_halo_from(p, 0, dst_dev, stream); }

int fem_plan_state_buffer(const fem_plan *p, int which, uint64_t *dev_ptr) {
  if (!p || !dev_ptr || (which != 0 && which != 1)) return fail(FEM_ERR_INVALID, "fem_plan_state_buffer: bad argument");
  *dev_ptr = (uint64_t)p->state[p->cur ^ which];
  return FEM_OK;
}

int fem_plan_pack_halo_from(fem_plan *p, int which, uint64_t dst_dev, void *stream) {
  if (!p || (which != 0 && which != 1)) return fail(FEM_ERR_INVALID, "fem_plan_pack_halo: bad argument");
  const void *from = p->state[p->cur ^ which];
  const fem_uint n = (fem_uint)p->send_idx.n;
  if (n == 0) return FEM_OK;
  if (!dst_dev) return fail(FEM_ERR_INVALID, "fem_plan_pack_halo: null destination");
  if (p->precision == FEM_PREC_F32)
    hipLaunchKernelGGL(pack_halo_kernel<float2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float2 *)from, p->send_idx.p, (float2 *)dst_dev, n);
  else
    hipLaunchKernelGGL(pack_halo_kernel<double2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const double2 *)from, p->send_idx.p, (double2 *)dst_dev, n);
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

int fem_plan_set_state(fem_plan *p, const double *nFn, const double *cFn) {
  return fem::guarded("fem_plan_set_state", [&]() -> int {
  if (!p || !nFn || !cFn) return fail(FEM_ERR_INVALID, "fem_plan_set_state: null");
  HIP_TRY(hipDeviceSynchronize());
  if (p->precision == FEM_PREC_F32) {
    std::vector<float> h(2 * (size_t)p->V);
    for (size_t i = 0; i < p->V; i++) { h[2 * i] = (float)(nFn[i] - p->base_n); h[2 * i + 1] = (float)(cFn[i] - p->base_c); }
    HIP_TRY(hipMemcpy(p->state[p->cur], h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  } else {
    std::vector<double> h(2 * (size_t)p->V);
    for (size_t i = 0; i < p->V; i++) { h[2 * i] = nFn[i] - p->base_n; h[2 * i + 1] = cFn[i] - p->base_c; }
    HIP_TRY(hipMemcpy(p->state[p->cur], h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
  }
  return FEM_OK;
  });
}

int fem_plan_get_state(fem_plan *p, double *nFn, double *cFn) {
  return fem::guarded("fem_plan_get_state", [&]() -> int {
  if (!p || !nFn || !cFn) return fail(FEM_ERR_INVALID, "fem_plan_get_state: null");
  HIP_TRY(hipDeviceSynchronize());
  if (p->persist > 0 && p->pctl.p) {            // the experimental persistent kernel gives up instead of hanging
    fem_uint gaveUp = 0;
    HIP_TRY(hipMemcpy(&gaveUp, p->pctl.p + 128, sizeof gaveUp, hipMemcpyDeviceToHost));
    if (gaveUp) return fail(FEM_ERR_DEVICE, ("tile_step_persist_kernel: gave up waiting for a neighbouring tile at item " + std::to_string(gaveUp - 1)).c_str());
  }
  if (p->precision == FEM_PREC_F32) {
    std::vector<float> h(2 * (size_t)p->V);
    HIP_TRY(hipMemcpy(h.data(), p->state[p->cur], sizeof(float) * h.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < p->V; i++) { nFn[i] = p->base_n + (double)h[2 * i]; cFn[i] = p->base_c + (double)h[2 * i + 1]; }
  } else {
    std::vector<double> h(2 * (size_t)p->V);
    HIP_TRY(hipMemcpy(h.data(), p->state[p->cur], sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < p->V; i++) { nFn[i] = p->base_n + h[2 * i]; cFn[i] = p->base_c + h[2 * i + 1]; }
  }
  return FEM_OK;
  });
}

static int launchTiles(fem_plan *p, fem_uint first, fem_uint count, int nsteps, double dt, const fem_model *model,
                       void *stream) {
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  ModelT<float> mf = {(float)(model->nMax - p->base_n), (float)model->D, (float)model->r, (float)model->alpha,
                      (float)model->S, (float)dt, (float)p->base_n, (float)(1.0 / (1.0 + p->base_n))};
  ModelT<double> md = {model->nMax - p->base_n, model->D, model->r, model->alpha, model->S, dt, p->base_n,
                       1.0 / (1.0 + p->base_n)};
  void *src = p->state[p->cur], *dst = p->state[p->cur ^ 1];
  hipError_t e = hipSuccess;
#define FEM_TILE_CASE(C)                                                                                       \
  case C:                                                                                                      \
    e = p->precision == FEM_PREC_F32 ? launchTile<float, C>(p, first, count, src, dst, nsteps, mf, (hipStream_t)stream) \
                                     : launchTile<double, C>(p, first, count, src, dst, nsteps, md, (hipStream_t)stream); \
    break;
  switch (p->tcfg) { FEM_TILE_CASE(0) FEM_TILE_CASE(1) FEM_TILE_CASE(2) FEM_TILE_CASE(3) FEM_TILE_CASE(4) FEM_TILE_CASE(5) FEM_TILE_CASE(6) FEM_TILE_CASE(7) FEM_TILE_CASE(8) }
#undef FEM_TILE_CASE
  if (e != hipSuccess) return fail(FEM_ERR_DEVICE, "tile_step_kernel launch", e);
  return FEM_OK;
}

int fem_plan_launch_tiles(fem_plan *p, fem_uint first, fem_uint count, int nsteps, double dt, const fem_model *model,
                          void *stream) {
  if (!p || p->ksteps <= 0) return fail(FEM_ERR_INVALID, "fem_plan_launch_tiles: not a tiled plan");
  if (nsteps < 1 || nsteps > p->ksteps || first > p->ntiles || count > p->ntiles - first)
    return fail(FEM_ERR_INVALID, "fem_plan_launch_tiles: tile range or step count out of range");
  const int rc = launchTiles(p, first, count, nsteps, dt, model, stream);
  if (rc != FEM_OK) return rc;
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

int fem_plan_flip(fem_plan *p) {
  if (!p) return fail(FEM_ERR_INVALID, "fem_plan_flip: null");
  p->cur ^= 1;
  return FEM_OK;
}

int fem_plan_persist_groups(const fem_plan *p, int *groups, int *min_groups) {
  if (!p || !groups) return fail(FEM_ERR_INVALID, "fem_plan_persist_groups: null");
  *groups = (p->ksteps > 0 && p->persist > 0 && p->packed && p->precision == FEM_PREC_F32 && p->pctl.p &&
             (p->tcfg == kDefaultTileCfg || p->tcfg == kDefaultTileCfgWide)) ? p->persist : 0;
  if (min_groups) *min_groups = *groups > 0 ? p->persist_min : 0;
  return FEM_OK;
}

int fem_plan_tile_classes(const fem_plan *p, fem_uint *n_interior, fem_uint *n_boundary, fem_uint *n_ghost) {
  if (!p || p->ksteps <= 0) return fail(FEM_ERR_INVALID, "fem_plan_tile_classes: not a tiled plan");
  if (n_interior) *n_interior = p->nInterior;
  if (n_boundary) *n_boundary = p->nBoundary;
  if (n_ghost) *n_ghost = p->nGhost;
  return FEM_OK;
}

int fem_plan_run(fem_plan *p, int nsteps, double dt, const fem_model *model, void *stream) {
  if (!p || nsteps < 0) return fail(FEM_ERR_INVALID, "fem_plan_run: bad argument");
  if (p->ksteps > 0 && p->persist > 0 && p->packed && p->precision == FEM_PREC_F32 &&
      (p->tcfg == kDefaultTileCfg || p->tcfg == kDefaultTileCfgWide)) {      // several groups per launch (persistRule)
    fem_model dm;
    if (!model) { fem_default_model(&dm); model = &dm; }
    const ModelT<float> mf = {(float)(model->nMax - p->base_n), (float)model->D, (float)model->r, (float)model->alpha,
                              (float)model->S, (float)dt, (float)p->base_n, (float)(1.0 / (1.0 + p->base_n))};
    for (int done = 0; done < nsteps;) {
      const int left = nsteps - done;
      const fem_uint groups = (fem_uint)std::min(p->persist, (left + p->ksteps - 1) / p->ksteps);
      if ((int)groups < p->persist_min) {       // too few groups to pay for a persistent launch: one launch per group
        for (int d2 = 0; d2 < left;) {
          const int c = std::min(p->ksteps, left - d2);
          const int rc = launchTiles(p, 0, p->ntiles, c, dt, model, stream);
          if (rc != FEM_OK) return rc;
          p->cur ^= 1;
          d2 += c;
        }
        break;
      }
      const int steps = std::min(left, (int)groups * p->ksteps);
      const int lastSteps = steps - ((int)groups - 1) * p->ksteps;
      const hipError_t e = p->tcfg == kDefaultTileCfg
          ? launchTilePersist<kDefaultTileCfg>(p, p->state[p->cur], p->state[p->cur ^ 1], p->ksteps, lastSteps, groups, mf, (hipStream_t)stream)
          : launchTilePersist<kDefaultTileCfgWide>(p, p->state[p->cur], p->state[p->cur ^ 1], p->ksteps, lastSteps, groups, mf, (hipStream_t)stream);
      if (e != hipSuccess) return fail(FEM_ERR_DEVICE, "tile_step_persist_kernel launch", e);
      p->cur ^= (int)(groups & 1u);
      done += steps;
    }
    HIP_TRY(hipGetLastError());
    return FEM_OK;
  }
  if (p->ksteps > 0) {                       // fused tiles: up to ksteps time steps per launch
    for (int done = 0; done < nsteps;) {
      const int c = std::min(p->ksteps, nsteps - done);
      const int rc = launchTiles(p, 0, p->ntiles, c, dt, model, stream);
      if (rc != FEM_OK) return rc;
      p->cur ^= 1;
      done += c;
    }
    HIP_TRY(hipGetLastError());
    return FEM_OK;
  }
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  ModelT<float> mf = {(float)(model->nMax - p->base_n), (float)model->D, (float)model->r, (float)model->alpha,
                      (float)model->S, (float)dt, (float)p->base_n, (float)(1.0 / (1.0 + p->base_n))};
  ModelT<double> md = {model->nMax - p->base_n, model->D, model->r, model->alpha, model->S, dt, p->base_n,
                       1.0 / (1.0 + p->base_n)};
  for (int i = 0; i < nsteps; i++) {
    void *src = p->state[p->cur], *dst = p->state[p->cur ^ 1];
    const bool nt = p->sell_nt >= 0 ? p->sell_nt != 0 : fem_plan_operator_bytes(p) > (256ull << 20);   // see launchTile
    if (p->precision == FEM_PREC_F32 && p->packed)
      hipLaunchKernelGGL((nt ? sell_step_kernel<float, true, false, true> : sell_step_kernel<float, false, false, true>), dim3(p->nblocks), dim3(256), 0,
                         (hipStream_t)stream, (const float2 *)src, (float2 *)dst, p->slice_off.p, p->idx.p, (const float4 *)p->coef,
                         p->V, p->nslices, p->nblocks, mf, (const float *)p->rscale);
    else if (p->precision == FEM_PREC_F32)
      hipLaunchKernelGGL((nt ? sell_step_kernel<float, true> : sell_step_kernel<float, false>), dim3(p->nblocks), dim3(256), 0, (hipStream_t)stream,
                         (const float2 *)src, (float2 *)dst, p->slice_off.p, p->idx.p, (const float4 *)p->coef,
                         p->V, p->nslices, p->nblocks, mf, (const float *)nullptr);
    else
      hipLaunchKernelGGL((nt ? sell_step_kernel<double, true> : sell_step_kernel<double, false>), dim3(p->nblocks), dim3(256), 0, (hipStream_t)stream,
                         (const double2 *)src, (double2 *)dst, p->slice_off.p, p->idx.p, (const double4 *)p->coef,
                         p->V, p->nslices, p->nblocks, md, (const float *)nullptr);
    p->cur ^= 1;
  }
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

// ---- semi-implicit stepping (runImex above) ---------------------------------------------------

int fem_plan_run_imex(fem_plan *p, int nsteps, double dt, const fem_model *model, int cg_iterations, void *stream) {
  return fem::guarded("fem_plan_run_imex", [&]() -> int {
  if (!p || nsteps < 0 || cg_iterations < 1 || cg_iterations > 10000) return fail(FEM_ERR_INVALID, "fem_plan_run_imex: bad argument");
  if (!p->can_imex) return fail(FEM_ERR_INVALID, "fem_plan_run_imex: plan was not created with FEM_PLAN_IMPLICIT");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  return p->precision == FEM_PREC_F32 ? runImex<float>(p, nsteps, dt, model, cg_iterations, (hipStream_t)stream)
                                      : runImex<double>(p, nsteps, dt, model, cg_iterations, (hipStream_t)stream);
  });
}

int fem_plan_run_imex_fused(fem_plan *p, int nsteps, double dt, const fem_model *model, double rel_tol, void *stream) {
  return fem::guarded("fem_plan_run_imex_fused", [&]() -> int {
  if (!p || nsteps < 0 || !(dt > 0) || !(rel_tol > 0) || rel_tol >= 1) return fail(FEM_ERR_INVALID, "fem_plan_run_imex_fused: bad argument (0 < rel_tol < 1)");
  if (p->ksteps < 1 || (p->tcfg != kDefaultTileCfg && p->tcfg != kDefaultTileCfgWide) || p->can_assemble)
    return fail(FEM_ERR_INVALID, "fem_plan_run_imex_fused: needs a fused-tile plan (fem_plan_create_tiled) on a default register tile, operator from the host assembly");
  if (p->nGhost || p->nBoundary) return fail(FEM_ERR_INVALID, "fem_plan_run_imex_fused: whole-mesh plans only");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  hipStream_t st = (hipStream_t)stream;
  if (p->precision == FEM_PREC_F32)
    return p->tcfg == kDefaultTileCfg ? runImexFused<float, kDefaultTileCfg>(p, nsteps, dt, model, rel_tol, st)
                                      : runImexFused<float, kDefaultTileCfgWide>(p, nsteps, dt, model, rel_tol, st);
  return p->tcfg == kDefaultTileCfg ? runImexFused<double, kDefaultTileCfg>(p, nsteps, dt, model, rel_tol, st)
                                    : runImexFused<double, kDefaultTileCfgWide>(p, nsteps, dt, model, rel_tol, st);
  });
}

int fem_plan_run_imex_fused_sharded(fem_plan *p, int nsteps, double dt, const fem_model *model, double rel_tol, double wsum_max,
                                    const fem_imex_comm *comm, void *stream) {
  return fem::guarded("fem_plan_run_imex_fused_sharded", [&]() -> int {
  if (!p || nsteps < 0 || !(dt > 0) || !(rel_tol > 0) || rel_tol >= 1 || !comm || !comm->exchange || !(wsum_max > 0))
    return fail(FEM_ERR_INVALID, "fem_plan_run_imex_fused_sharded: bad argument (0 < rel_tol < 1, wsum_max > 0, an exchange callback)");
  if (p->ksteps < 1 || (p->tcfg != kDefaultTileCfg && p->tcfg != kDefaultTileCfgWide) || p->can_assemble)
    return fail(FEM_ERR_INVALID, "fem_plan_run_imex_fused_sharded: needs a shard's fused-tile plan on a default register tile, operator from the host assembly");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  hipStream_t st = (hipStream_t)stream;
  if (p->precision == FEM_PREC_F32)
    return p->tcfg == kDefaultTileCfg ? runImexFused<float, kDefaultTileCfg>(p, nsteps, dt, model, rel_tol, st, comm, wsum_max)
                                      : runImexFused<float, kDefaultTileCfgWide>(p, nsteps, dt, model, rel_tol, st, comm, wsum_max);
  return p->tcfg == kDefaultTileCfg ? runImexFused<double, kDefaultTileCfg>(p, nsteps, dt, model, rel_tol, st, comm, wsum_max)
                                    : runImexFused<double, kDefaultTileCfgWide>(p, nsteps, dt, model, rel_tol, st, comm, wsum_max);
  });
}

int fem_plan_imex_fused_iterations(const fem_plan *p) { return p ? p->cheb_iters : 0; }

int fem_plan_run_imex_sharded(fem_plan *p, int nsteps, double dt, const fem_model *model, int cg_iterations, fem_uint n_own,
                              const fem_imex_comm *comm, void *stream) {
  return fem::guarded("fem_plan_run_imex_sharded", [&]() -> int {
  if (!p || nsteps < 0 || cg_iterations < 1 || cg_iterations > 10000 || !comm || !comm->exchange || !comm->allreduce_sum || n_own > p->V)
    return fail(FEM_ERR_INVALID, "fem_plan_run_imex_sharded: bad argument");
  if (!p->can_imex) return fail(FEM_ERR_INVALID, "fem_plan_run_imex_sharded: plan was not created with FEM_PLAN_IMPLICIT");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  return p->precision == FEM_PREC_F32 ? runImex<float>(p, nsteps, dt, model, cg_iterations, (hipStream_t)stream, comm, n_own)
                                      : runImex<double>(p, nsteps, dt, model, cg_iterations, (hipStream_t)stream, comm, n_own);
  });
}

int fem_plan_pack_vector(fem_plan *p, uint64_t dev_vector, uint64_t dst_dev, void *stream) {
  if (!p || !dev_vector) return fail(FEM_ERR_INVALID, "fem_plan_pack_vector: bad argument");
  const fem_uint n = (fem_uint)p->send_idx.n;
  if (n == 0) return FEM_OK;
  if (!dst_dev) return fail(FEM_ERR_INVALID, "fem_plan_pack_vector: null destination");
  if (p->precision == FEM_PREC_F32)
    hipLaunchKernelGGL(pack_halo_kernel<float2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float2 *)dev_vector,
                       p->send_idx.p, (float2 *)dst_dev, n);
  else
    hipLaunchKernelGGL(pack_halo_kernel<double2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double2 *)dev_vector,
                       p->send_idx.p, (double2 *)dst_dev, n);
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

int fem_plan_imex_tolerance(fem_plan *p, double rel_tol) {
  if (!p || !p->can_imex || !(rel_tol >= 0) || rel_tol >= 1) return fail(FEM_ERR_INVALID, "fem_plan_imex_tolerance: plan without FEM_PLAN_IMPLICIT, or tolerance outside [0, 1)");
  p->imex_tol = rel_tol;
  return FEM_OK;
}

int fem_plan_imex_iterations(fem_plan *p, int *iterations) {
  if (!p || !p->ix_sc || !iterations) return fail(FEM_ERR_INVALID, "fem_plan_imex_iterations: no semi-implicit step has run");
  double sc[16];
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(sc, p->ix_sc, sizeof sc, hipMemcpyDeviceToHost));
  *iterations = (int)sc[11];
  return FEM_OK;
}

int fem_plan_imex_residual(fem_plan *p, double *rel_n, double *rel_c) {
  if (!p || !p->ix_sc) return fail(FEM_ERR_INVALID, "fem_plan_imex_residual: no semi-implicit step has run");
  double sc[16];
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(sc, p->ix_sc, sizeof sc, hipMemcpyDeviceToHost));
  if (rel_n) *rel_n = sc[6] > 0 ? std::sqrt(sc[8] / sc[6]) : 0.0;
  if (rel_c) *rel_c = sc[7] > 0 ? std::sqrt(sc[9] / sc[7]) : 0.0;
  return FEM_OK;
}

int fem_plan_state(const fem_plan *p, uint64_t *dev_ptr, uint64_t *n_vertices, int *scalar_bytes,
                   double *base_n, double *base_c) {
  if (!p) return fail(FEM_ERR_INVALID, "fem_plan_state: null");
  if (base_n) *base_n = p->base_n;
  if (base_c) *base_c = p->base_c;
  if (dev_ptr) *dev_ptr = (uint64_t)p->state[p->cur];
  if (n_vertices) *n_vertices = p->V;
  if (scalar_bytes) *scalar_bytes = p->precision == FEM_PREC_F32 ? 4 : 8;
  return FEM_OK;
}


// ---- device assembly -----------------------------------------------------------------------------

int fem_plan_set_points(fem_plan *p, const double *points) {
  return fem::guarded("fem_plan_set_points", [&]() -> int {
  if (!p || !points || !p->can_assemble) return fail(FEM_ERR_INVALID, "fem_plan_set_points: plan was not created with FEM_PLAN_DEVICE_ASSEMBLY");
  return p->precision == FEM_PREC_F32 ? setPoints<float>(p, points) : setPoints<double>(p, points);
  });
}

int fem_plan_assemble(fem_plan *p, void *stream) {
  if (!p || !p->can_assemble) return fail(FEM_ERR_INVALID, "fem_plan_assemble: plan was not created with FEM_PLAN_DEVICE_ASSEMBLY");
  if (p->precision == FEM_PREC_F32) launchAssemble<float>(p, (hipStream_t)stream);
  else launchAssemble<double>(p, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

int fem_plan_assemble_step(fem_plan *p, double dt, const fem_model *model, int store_operator, void *stream) {
  if (!p || !p->can_assemble) return fail(FEM_ERR_INVALID, "fem_plan_assemble_step: plan was not created with FEM_PLAN_DEVICE_ASSEMBLY");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  // the fused kernel: sliced-ELL plans whose rows are fans of at most 12 entries; anything else takes the two
  // kernels one after the other (same result: the fused kernel applies the coefficients it would have stored)
  if (p->ksteps == 0 && p->fan_rows && p->max_row <= 12) {
    if (p->precision == FEM_PREC_F32) launchAssembleStep<float>(p, dt, model, store_operator != 0, (hipStream_t)stream);
    else launchAssembleStep<double>(p, dt, model, store_operator != 0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return FEM_OK;
  }
  const int rc = fem_plan_assemble(p, stream);
  return rc != FEM_OK ? rc : fem_plan_run(p, 1, dt, model, stream);
}

int fem_plan_deform(fem_plan *p, double amplitude, double wavenumber, double phase, void *stream) {
  if (!p || !p->can_assemble) return fail(FEM_ERR_INVALID, "fem_plan_deform: plan was not created with FEM_PLAN_DEVICE_ASSEMBLY");
  if (!p->rows_are_vertices) {           // a shard's plan: only the vertices its rows stand for
    const dim3 rgrid((p->V + 255) / 256);
    if (p->precision == FEM_PREC_F32)
      hipLaunchKernelGGL(deform_rows_kernel<float>, rgrid, dim3(256), 0, (hipStream_t)stream, (const float4 *)p->pos0, (float4 *)p->pos,
                         p->rowg.p, p->V, (float)amplitude, (float)wavenumber, (float)phase);
    else
      hipLaunchKernelGGL(deform_rows_kernel<double>, rgrid, dim3(256), 0, (hipStream_t)stream, (const double4 *)p->pos0, (double4 *)p->pos,
                         p->rowg.p, p->V, amplitude, wavenumber, phase);
    HIP_TRY(hipGetLastError());
    return FEM_OK;
  }
  const dim3 grid((p->Vg + 255) / 256);
  if (p->precision == FEM_PREC_F32)
    hipLaunchKernelGGL(deform_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)p->pos0, (float4 *)p->pos,
                       p->Vg, (float)amplitude, (float)wavenumber, (float)phase);
  else
    hipLaunchKernelGGL(deform_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double4 *)p->pos0, (double4 *)p->pos,
                       p->Vg, amplitude, wavenumber, phase);
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

int fem_plan_max_edge(fem_plan *p, double *maxEdgeLen) {
  if (!p || !p->can_assemble || !maxEdgeLen) return fail(FEM_ERR_INVALID, "fem_plan_max_edge: bad argument");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(maxEdgeLen, p->maxedge, 8, hipMemcpyDeviceToHost));
  return FEM_OK;
}

int fem_plan_face_records(fem_plan *p, double *rec16F) {
  return fem::guarded("fem_plan_face_records", [&]() -> int {
  if (!p || !p->can_assemble || !rec16F) return fail(FEM_ERR_INVALID, "fem_plan_face_records: bad argument");
  HIP_TRY(p->precision == FEM_PREC_F32 ? launchFaceRecords<float>(p) : launchFaceRecords<double>(p));
  return p->precision == FEM_PREC_F32 ? downloadAs<float>(p->rec, 16 * (size_t)p->F, rec16F)
                                      : downloadAs<double>(p->rec, 16 * (size_t)p->F, rec16F);
  });
}

uint64_t fem_plan_coef_count(const fem_plan *p) {
  if (!p) return 0;
  return p->ksteps > 0 ? p->tile_entries * 4 : (uint64_t)p->entries * 4;
}

int fem_plan_download_coef(fem_plan *p, double *out) {
  return fem::guarded("fem_plan_download_coef", [&]() -> int {
  if (!p || !out) return fail(FEM_ERR_INVALID, "fem_plan_download_coef: null");
  const void *dev = p->ksteps > 0 ? p->tcoef : p->coef;
  const size_t n = fem_plan_coef_count(p);
  if (p->packed && p->ksteps == 0) {   // sliced ELL: records at 3 * e words, scales per row
    const size_t entries = n / 4;
    std::vector<fem_uint> rec(3 * entries), soff(p->nslices + 1);
    std::vector<float> scale(p->V);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rec.data(), dev, 4 * rec.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(scale.data(), p->rscale, 4 * scale.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(soff.data(), p->slice_off.p, 4 * soff.size(), hipMemcpyDeviceToHost));
    for (size_t sl = 0; sl < p->nslices; sl++)
      for (size_t e = soff[sl]; e < soff[sl + 1]; e++) {
        const size_t v = std::min<size_t>(64 * sl + (e - soff[sl]) % 64, p->V - 1);
        const fem_uint a = rec[3 * e + 1], b = rec[3 * e + 2];
        float wt;
        memcpy(&wt, &rec[3 * e], 4);
        const float sc = scale[v];
        const int qx = (int)(a << 11) >> 11, qy = (int)(((a >> 21) | (b << 11)) << 11) >> 11, qz = (int)(b << 1) >> 11;
        out[4 * e] = wt; out[4 * e + 1] = (float)qx * sc; out[4 * e + 2] = (float)qy * sc; out[4 * e + 3] = (float)qz * sc;
      }
    return FEM_OK;
  }
  if (p->packed) {                     // 12-byte records: decoded the way the kernel decodes them
    std::vector<float> c32(n);
    std::vector<fem_uint> meta(16 * (size_t)p->ntiles);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(c32.data(), dev, 4 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(meta.data(), p->tmeta.p, 4 * meta.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) out[i] = c32[i];       // extras of wide rows: four floats as they are
    const fem_uint *words = reinterpret_cast<const fem_uint *>(c32.data());
    for (size_t t = 0; t < p->ntiles; t++) {
      const size_t roff = meta[16 * t + 2], rpad = meta[16 * t + 3];
      const fem_uint *block = words + 4 * roff;
      for (size_t w = 0; w < (size_t)kTileW; w++)
        for (size_t r = 0; r < rpad; r++) {
          const size_t e = roff + w * rpad + r, le = w * rpad + r;
          const fem_uint a = block[3 * le + 1], b = block[3 * le + 2];
          float wt, sc;
          memcpy(&wt, &block[3 * le], 4);
          memcpy(&sc, &block[3 * (size_t)kTileW * rpad + r], 4);
          const int qx = (int)(a << 11) >> 11, qy = (int)(((a >> 21) | (b << 11)) << 11) >> 11, qz = (int)(b << 1) >> 11;
          out[4 * e] = wt; out[4 * e + 1] = (float)qx * sc; out[4 * e + 2] = (float)qy * sc; out[4 * e + 3] = (float)qz * sc;
        }
    }
    return FEM_OK;
  }
  if (p->mixed) {                      // fp64 weights + fp32 gradient coefficients, widened
    std::vector<double> raw(n);
    std::vector<fem_uint> meta(16 * (size_t)p->ntiles);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(raw.data(), dev, 8 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(meta.data(), p->tmeta.p, 4 * meta.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) out[i] = raw[i];       // extras of wide rows: four doubles as they are
    for (size_t t = 0; t < p->ntiles; t++) {
      const size_t roff = meta[16 * t + 2], rpad = meta[16 * t + 3], cnt = (size_t)kTileW * rpad;
      const double *block = raw.data() + 4 * roff;
      const unsigned char *g = reinterpret_cast<const unsigned char *>(block + cnt);
      for (size_t le = 0; le < cnt; le++) {
        float c3[3];
        memcpy(c3, g + 12 * le, 12);
        out[4 * (roff + le)] = block[le];
        for (int d = 0; d < 3; d++) out[4 * (roff + le) + 1 + d] = c3[d];
      }
    }
    return FEM_OK;
  }
  return p->precision == FEM_PREC_F32 ? downloadAs<float>(dev, n, out) : downloadAs<double>(dev, n, out);
  });
}


// ---- result output ------------------------------------------------------------------------------------

int fem_plan_format(fem_plan *p, fem_uint nverts, double *rgba, unsigned char *rgb, double *nmin, double *nmax) {
  return fem::guarded("fem_plan_format", [&]() -> int {
  if (!p || nverts == 0 || nverts > p->V) return fail(FEM_ERR_INVALID, "fem_plan_format: bad vertex count");
  const fem_uint V = nverts, nb = std::min<fem_uint>((V + 255) / 256, 1024);
  double *part = nullptr, *mm = nullptr;
  double4 *out = nullptr;
  hipError_t e = hipDeviceSynchronize();      // steps may still be in flight on a non-blocking stream
  if (e == hipSuccess) e = hipMalloc((void **)&part, 16 * (size_t)nb);
  if (e == hipSuccess) e = hipMalloc((void **)&mm, 16);
  if (e == hipSuccess) e = hipMalloc((void **)&out, sizeof(double4) * (size_t)V);
  if (e == hipSuccess) {
    if (p->precision == FEM_PREC_F32) hipLaunchKernelGGL(minmax_kernel<float2>, dim3(nb), dim3(256), 0, 0, (const float2 *)p->state[p->cur], V, part);
    else hipLaunchKernelGGL(minmax_kernel<double2>, dim3(nb), dim3(256), 0, 0, (const double2 *)p->state[p->cur], V, part);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, 0, (const double *)part, nb, mm);
    if (p->precision == FEM_PREC_F32) hipLaunchKernelGGL(format_kernel<float2>, dim3((V + 255) / 256), dim3(256), 0, 0, (const float2 *)p->state[p->cur], V, (const double *)mm, out);
    else hipLaunchKernelGGL(format_kernel<double2>, dim3((V + 255) / 256), dim3(256), 0, 0, (const double2 *)p->state[p->cur], V, (const double *)mm, out);
    e = hipDeviceSynchronize();
  }
  std::vector<double> h;
  double hmm[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(hmm, mm, 16, hipMemcpyDeviceToHost);
  if (e == hipSuccess && (rgba || rgb)) {
    h.resize(4 * (size_t)V);
    e = hipMemcpy(h.data(), out, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
  }
  for (void *q : {(void *)part, (void *)mm, (void *)out}) if (q) (void)hipFree(q);
  if (e != hipSuccess) return fail(FEM_ERR_DEVICE, "fem_plan_format", e);
  if (rgba) memcpy(rgba, h.data(), sizeof(double) * h.size());
  if (rgb)      // 8-bit colours as in ResultantPattern.off: components truncated, so R + G = 254 or 255
    for (size_t i = 0; i < V; i++)
      for (int k = 0; k < 3; k++) {
        const double c = 255.0 * h[4 * i + k];
        rgb[3 * i + k] = (unsigned char)(c > 0 ? (c > 255 ? 255 : c) : 0);      // NaN and negatives -> 0
      }
  if (nmin) *nmin = hmm[0] + p->base_n;     // the buffers hold deviations from base_n
  if (nmax) *nmax = hmm[1] + p->base_n;
  return FEM_OK;
  });
}

// ---- result output without allocation ---------------------------------------------------------------
// fem_plan_format above allocates its partial-extrema buffer, the extrema pair and a staging buffer on every
// call.  Callers that cannot allocate (HIP graph capture, tight serving loops) get the same three kernels with the
// partials and the extrema in a file-scope device array and the colours written straight into the caller's
// device buffer.  The array exists once per GPU (each device holds its own copy at its own address, looked up
// once per device with that device current); every call on a GPU shares it, so calls that overlap on one GPU race.
constexpr fem_uint kFormatMaxBlocks = 1024;                        // partial (min, max) pairs
constexpr fem_uint kFormatNoAllocMaxVerts = 1u << 26;              // 67 108 864 vertices: 256 grid-stride trips of 1024 x 256 threads
static __device__ __align__(256) double scratch[2 * kFormatMaxBlocks + 2];
constexpr int kMaxDevices = 64;
static double *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mutex;

static int formatScratch(const fem_plan *p, double **out) {
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  // the kernels run on the current device against that device's copy of the array: the plan's buffers must live there
  if (p->device >= 0 && dev != p->device)
    return fail(FEM_ERR_INVALID, ("fem_plan_format_noalloc: the plan was created on GPU " + std::to_string(p->device) +
                                  " but GPU " + std::to_string(dev) + " is current (hipSetDevice to the plan's device first)").c_str());
  if (dev < 0 || dev >= kMaxDevices) return fail(FEM_ERR_LIMIT, "fem_plan_format_noalloc: device ordinal beyond 63");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    double *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));      // this device's copy of the array
    if (!ptr) return fail(FEM_ERR_DEVICE, "fem_plan_format_noalloc: no device address for the scratch array");
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return FEM_OK;
}

int fem_plan_format_noalloc(fem_plan *p, fem_uint nverts, uint64_t dev_rgba, uint64_t dev_minmax, void *stream) {
  if (!p || nverts == 0 || nverts > p->V) return fail(FEM_ERR_INVALID, "fem_plan_format_noalloc: bad vertex count");
  if (nverts > kFormatNoAllocMaxVerts)
    return fail(FEM_ERR_LIMIT, "fem_plan_format_noalloc: more than 67108864 vertices (the fixed scratch array is sized for that many); use fem_plan_format");
  if (!dev_rgba || (dev_rgba & 31u)) return fail(FEM_ERR_INVALID, "fem_plan_format_noalloc: dev_rgba must be a 32-byte aligned device buffer of 4 * nverts doubles");
  if (dev_minmax & 7u) return fail(FEM_ERR_INVALID, "fem_plan_format_noalloc: dev_minmax must be 8-byte aligned");
  double *ws = nullptr;
  const int rc = formatScratch(p, &ws);
  if (rc != FEM_OK) return rc;
  double *part = ws, *mm = ws + 2 * kFormatMaxBlocks;
  const fem_uint V = nverts, nb = std::min<fem_uint>((V + 255) / 256, kFormatMaxBlocks);
  hipStream_t st = (hipStream_t)stream;
  double4 *out = (double4 *)dev_rgba;
  (void)hipGetLastError();        // an earlier call's error is not this call's: only what these launches raise is reported
  // every block writes its own partial pair and the final kernel reads exactly nb of them: nothing in the array
  // has to start at zero, whatever the previous call left there
  if (p->precision == FEM_PREC_F32) hipLaunchKernelGGL(minmax_kernel<float2>, dim3(nb), dim3(256), 0, st, (const float2 *)p->state[p->cur], V, part);
  else hipLaunchKernelGGL(minmax_kernel<double2>, dim3(nb), dim3(256), 0, st, (const double2 *)p->state[p->cur], V, part);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, st, (const double *)part, nb, mm);
  if (p->precision == FEM_PREC_F32) hipLaunchKernelGGL(format_kernel<float2>, dim3((V + 255) / 256), dim3(256), 0, st, (const float2 *)p->state[p->cur], V, (const double *)mm, out);
  else hipLaunchKernelGGL(format_kernel<double2>, dim3((V + 255) / 256), dim3(256), 0, st, (const double2 *)p->state[p->cur], V, (const double *)mm, out);
  if (dev_minmax)
    hipLaunchKernelGGL(minmax_export_kernel, dim3(1), dim3(64), 0, st, (const double *)mm, p->base_n, (double *)dev_minmax);
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

int fem_plan_tile_info(const fem_plan *p, int *ksteps, fem_uint *ntiles, fem_uint *max_local, uint64_t *rows_loaded,
                       int *threads, int *rows_per_thread) {
  if (!p) return fail(FEM_ERR_INVALID, "fem_plan_tile_info: null");
  if (ksteps) *ksteps = p->ksteps;
  if (ntiles) *ntiles = p->ntiles;
  if (max_local) *max_local = p->maxLocal;
  if (rows_loaded) *rows_loaded = p->tile_rows;
  if (threads) *threads = p->ksteps ? kTileCfgs[p->tcfg].tpb : 256;
  if (rows_per_thread) *rows_per_thread = p->ksteps ? kTileCfgs[p->tcfg].r : 1;
  return FEM_OK;
}

int fem_plan_operator_format(const fem_plan *p) { return !p ? 0 : (p->packed ? 1 : (p->mixed ? 2 : 0)); }

uint64_t fem_plan_operator_bytes(const fem_plan *p) {
  if (!p) return 0;
  const size_t sb = p->precision == FEM_PREC_F32 ? 4 : 8;
  if (p->ksteps > 0) return p->tile_bytes;
  if (p->packed) return (uint64_t)p->entries * (4 + 12) + (uint64_t)p->V * 4 + (uint64_t)(p->nslices + 1) * 4;
  return (uint64_t)p->entries * (4 + 4 * sb) + (uint64_t)(p->nslices + 1) * 4;
}

}  // extern "C"
