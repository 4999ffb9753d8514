// fem_host.hpp (private) -- the host side's internals: device layouts, packers, shards, worker threads.
// The reference-named public interfaces (SimpleTriMesh, MeshStats, initMeshStats, compute*CPU, partition, mesh IO)
// are declared in include/fem_host.hpp.
#pragma once

#include <cstdint>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/fem_host.hpp"

namespace fem {

// Reference device layout (initializeGPUData, main.cpp:207-342), dense host form.
struct RefLayout {
  int n = 0;
  uint max_vertex_count = 0, vv_max_neighbors = 0, max_face_count = 0;
  std::vector<uint> halo_faces_keys, halo_faces, block_face_count;
  std::vector<double> vertex_weights;        // [n][maxV]
  std::vector<uint> nbr_v;                   // [n][W+1][maxV]
  std::vector<double> vv_weights;            // [n][W][maxV]
  std::vector<uint> face_vertices;           // [n][3][maxF]
  std::vector<double> fv_weights_new;        // [n][3][maxF]
  std::vector<fem_float4> he_grads;          // [n][2][maxF]
};
int packRefLayout(const SimpleTriMesh &m, const MeshStats &s, int n, const uint *node_parts,
                  const uint *element_parts, RefLayout &out);

// Native layout: sliced ELL (64 rows per slice), per entry the neighbour id and
// (w, Cx, Cy, Cz): Laplacian weight and folded vertex-gradient coefficient.
struct SellLayout {
  uint V = 0, nslices = 0;
  std::vector<uint> slice_off;               // [nslices+1] entry offsets (multiples of 64)
  std::vector<uint> idx;                     // [entries]
  std::vector<double> coef;                  // [4*entries] (w, Cx, Cy, Cz)
  std::vector<uint> topo;                    // [2*entries] RowOp::nb in layout order, for device assembly
};
// Folded operator in CSR form over local rows (whole mesh, or a shard's loc2glob list).
struct RowOp {
  uint V = 0;
  std::vector<uint> rowptr, col;             // [V+1], [nnz] local column ids: ascending global order, or (fan = true) the
                                             // counter-clockwise order of the fan around the vertex
  bool fanOrdered = false;                   // every row is one fan in order: entry k's left face ends at entry k+1 (the
                                             // last one wraps to entry 0 on a closed fan), its right face at entry k-1
  std::vector<uint> nb;                      // [2*nnz] (mesh id of neighbour j, s1 | s2 << 8 | 1 << 16): s1/s2 = row slots of
                                             // the vertices opposite edge (v, j) in the faces left of v->j / j->v (0xff = none)
  std::vector<uint> rowg;                    // [V] mesh vertex id of each row
  std::vector<double> coef;                  // [4*nnz] (w, Cx, Cy, Cz)
};
int buildRowOp(const SimpleTriMesh &m, const MeshStats &s, RowOp &out, const uint *loc2glob = nullptr,
               uint nloc = 0, bool fan = false);
int packSellLayout(const RowOp &op, SellLayout &out);

// Tiles of the fused k-step kernel (pack.cpp).
struct TileLayout {
  int ksteps = 0;
  uint T = 0, R = 0, W = 0, ntiles = 0, maxLocal = 0, maxExtra = 0;
  uint64_t rowSteps = 0;                     // sum over tiles of m[k-1] (rows loaded per launch)
  std::vector<uint> meta;                    // [ntiles][16]
  std::vector<uint> ghost;                   // row ids of the ring rows of every tile
  std::vector<uint> idx2;                    // packed 16-bit local column ids
  std::vector<uint> xinfo;                   // rows wider than W: per-row (first | count << 16), then column ids
  std::vector<double> coef;                  // [4*entries]
  std::vector<uint> topo;                    // [2*entries] RowOp::nb in layout order, for device assembly
};
int packTileLayout(const RowOp &op, const std::vector<std::pair<uint, uint>> &ranges, int ksteps, uint T, uint R,
                   uint W, uint maxLocal, uint maxExtra, TileLayout &out);

// One rank's share of an ordered mesh: a contiguous range of partitions (hence of vertices),
// plus the ghost vertices within `krings` edges of it, and who sends what to whom.
struct Shard {
  int rank = 0, nranks = 1, krings = 1;
  uint own_begin = 0, own_end = 0;           // global (ordered) vertex range owned by this rank
  std::vector<uint> loc2glob;                // owned (ascending) then ghosts (ascending global id)
  std::vector<uint> send_idx;                // local ids to send, grouped by destination rank
  std::vector<int> send_counts, recv_counts; // [nranks]
};
int buildShard(const SimpleTriMesh &m, int n, const uint *node_parts, int rank, int nranks, int krings,
               Shard &out);

// host worker threads for the embarrassingly parallel loops (FEM_HOST_THREADS, default min(16, cores))
int hostThreads();
// fn(begin, end) over [0, n) in contiguous chunks, one per thread; serial below `grain` items
template <typename Fn> void parallelFor(size_t n, size_t grain, Fn fn) {
  const int T = n >= grain ? hostThreads() : 1;
  if (T <= 1) { fn((size_t)0, n); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)T);          // an exception in a worker (bad_alloc ...) is rethrown by the caller
  for (int t = 0; t < T; t++)
    th.emplace_back([=, &err]() {
      try { fn(n * (size_t)t / (size_t)T, n * (size_t)(t + 1) / (size_t)T); }
      catch (...) { err[(size_t)t] = std::current_exception(); }
    });
  for (auto &x : th) x.join();
  for (auto &e : err) if (e) std::rethrow_exception(e);
}

// rigorous upper bound on the largest eigenvalue of -L (assembled Laplacian rows of the stats); mesh_stats.cpp
double laplacianSpectrumBound(const SimpleTriMesh &m, const MeshStats &s);

// records status + message for fem_last_error()/fem_error_detail(); returns status
int fail(int status, const char *what);

// Body of an extern "C" entry point that allocates on the host: no exception crosses the C boundary
// (fem_hip.h: "none aborts") -- out of memory becomes FEM_ERR_ALLOC, anything else FEM_ERR_INVALID.
template <typename Fn> int guarded(const char *where, Fn fn) {
  try { return fn(); }
  catch (const std::bad_alloc &) { return fail(FEM_ERR_ALLOC, where); }
  catch (const std::exception &e) { return fail(FEM_ERR_INVALID, e.what()); }
  catch (...) { return fail(FEM_ERR_INVALID, where); }
}

}  // namespace fem

// opaque C handles
struct fem_mesh {
  fem::SimpleTriMesh mesh;
  fem::MeshStats stats;
};
struct fem_ref_layout {
  fem::RefLayout l;
};
struct fem_shard {
  fem::Shard s;
};
