// capi.cpp -- host half of the C-ABI declared in include/fem_hip.h.
#include <cstdlib>
#include <cstring>
#include <new>

#include "fem_host.hpp"

using fem::fail;

namespace {
template <typename T> T *dupArray(const std::vector<T> &v) {
  T *p = (T *)malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
  if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}
int exportMesh(const fem::SimpleTriMesh &m, int *V, int *F, double **points, fem_uint **tri) {
  *V = (int)m.n_vertices(); *F = (int)m.n_faces();
  *points = dupArray(m.points); *tri = dupArray(m.faces);
  if (!*points || !*tri) { free(*points); free(*tri); return fail(FEM_ERR_ALLOC, "out of memory"); }
  return FEM_OK;
}
}  // namespace

extern "C" {

void fem_default_model(fem_model *m) {      // reference src/FEM_common.cu:9-13
  m->nMax = 1; m->D = 0.25; m->r = 59.453790914308100; m->alpha = 269; m->S = 1;
}

const char *fem_error_string(int status) {
  switch (status) {
    case FEM_OK: return "ok";
    case FEM_ERR_INVALID: return "invalid argument";
    case FEM_ERR_LIMIT: return "partition exceeds a kernel limit";
    case FEM_ERR_DEVICE: return "HIP runtime error";
    case FEM_ERR_ALLOC: return "out of memory";
    case FEM_ERR_IO: return "file cannot be read/written or is not a valid OFF";
    case FEM_ERR_NONMANIFOLD: return "mesh is not an oriented manifold (a directed edge occurs twice)";
    default: return "unknown status";
  }
}

void fem_free(void *p) { free(p); }

int fem_off_read(const char *path, int *V, int *F, double **points, fem_uint **tri, unsigned char **rgb) {
  return fem::guarded("fem_off_read", [&]() -> int {
  if (!path || !V || !F || !points || !tri) return fail(FEM_ERR_INVALID, "fem_off_read: null");
  fem::SimpleTriMesh m;
  int rc = fem::readOFF(path, m);
  if (rc != FEM_OK) return fail(rc, path);
  rc = exportMesh(m, V, F, points, tri);
  if (rc == FEM_OK && rgb) *rgb = m.colors.empty() ? nullptr : dupArray(m.colors);
  return rc;
  });
}

int fem_ply_read(const char *path, int *V, int *F, double **points, fem_uint **tri, unsigned char **rgb) {
  return fem::guarded("fem_ply_read", [&]() -> int {
  if (!path || !V || !F || !points || !tri) return fail(FEM_ERR_INVALID, "fem_ply_read: null");
  fem::SimpleTriMesh m;
  int rc = fem::readPLY(path, m);
  if (rc != FEM_OK) return fail(rc, path);
  rc = exportMesh(m, V, F, points, tri);
  if (rc == FEM_OK && rgb) *rgb = m.colors.empty() ? nullptr : dupArray(m.colors);
  return rc;
  });
}

int fem_off_write(const char *path, int V, int F, const double *points, const fem_uint *tri,
                  const unsigned char *rgb) {
  return fem::guarded("fem_off_write", [&]() -> int {
  if (!path || V < 0 || F < 0 || (V && !points) || (F && !tri)) return fail(FEM_ERR_INVALID, "fem_off_write: bad argument");
  fem::SimpleTriMesh m;
  m.points.assign(points, points + 3 * (size_t)V);
  m.faces.assign(tri, tri + 3 * (size_t)F);
  if (rgb) m.colors.assign(rgb, rgb + 3 * (size_t)V);
  int rc = fem::writeOFF(path, m);
  return rc == FEM_OK ? rc : fail(rc, path);
  });
}

int fem_make_icosphere(int nu, double edge, int *V, int *F, double **points, fem_uint **tri) {
  return fem::guarded("fem_make_icosphere", [&]() -> int {
  fem::SimpleTriMesh m;
  int rc = fem::makeIcosphere(nu, edge, m);
  if (rc != FEM_OK) return fail(rc, "fem_make_icosphere: need 1 <= nu <= 4000, edge > 0");
  return exportMesh(m, V, F, points, tri);
  });
}

int fem_make_scanned_like(int nu, double edge, fem_uint seed, double flip_fraction, double noise_amp,
                          double high_valence_fraction, int high_valence, int *V, int *F, double **points,
                          fem_uint **tri) {
  return fem::guarded("fem_make_scanned_like", [&]() -> int {
  fem::SimpleTriMesh m;
  int rc = fem::makeScannedLike(nu, edge, seed, flip_fraction, noise_amp, high_valence_fraction, high_valence, m);
  if (rc != FEM_OK) return fail(rc, "fem_make_scanned_like: bad argument");
  return exportMesh(m, V, F, points, tri);
  });
}

int fem_make_strip(int nx, int ny, double h, int *V, int *F, double **points, fem_uint **tri) {
  return fem::guarded("fem_make_strip", [&]() -> int {
  fem::SimpleTriMesh m;
  int rc = fem::makeStrip(nx, ny, h, m);
  if (rc != FEM_OK) return fail(rc, "fem_make_strip: need nx, ny >= 2, h > 0");
  return exportMesh(m, V, F, points, tri);
  });
}

int fem_subdivide(int V, int F, const double *points, const fem_uint *tri, int levels, int *Vout, int *Fout,
                  double **points_out, fem_uint **tri_out) {
  return fem::guarded("fem_subdivide", [&]() -> int {
  if (V < 0 || F < 0 || (V && !points) || (F && !tri)) return fail(FEM_ERR_INVALID, "fem_subdivide: bad argument");
  fem::SimpleTriMesh in, out;
  in.points.assign(points, points + 3 * (size_t)V);
  in.faces.assign(tri, tri + 3 * (size_t)F);
  int rc = fem::subdivide(in, levels, out);
  if (rc != FEM_OK) return fail(rc, "fem_subdivide: 0 <= levels <= 12, manifold input, result below 2^32 vertices");
  return exportMesh(out, Vout, Fout, points_out, tri_out);
  });
}

void fem_initial_condition(int V, fem_uint seed, double nMax, double *nFn, double *cFn) {
  fem::initialCondition(V, seed, nMax, nFn, cFn);
}

// ---- mesh handle -------------------------------------------------------------

int fem_mesh_create(int V, int F, const double *points, const fem_uint *tri, fem_mesh **out) {
  return fem::guarded("fem_mesh_create", [&]() -> int {
  if (V <= 0 || F < 0 || !points || (F && !tri) || !out) return fail(FEM_ERR_INVALID, "fem_mesh_create: bad argument");
  fem_mesh *m = new (std::nothrow) fem_mesh();
  if (!m) return fail(FEM_ERR_ALLOC, "out of memory");
  m->mesh.points.assign(points, points + 3 * (size_t)V);
  m->mesh.faces.assign(tri, tri + 3 * (size_t)F);
  int rc = m->mesh.update_topology();
  if (rc != FEM_OK) { delete m; return fail(rc, "fem_mesh_create: bad connectivity"); }
  *out = m;
  return FEM_OK;
  });
}

void fem_mesh_destroy(fem_mesh *m) { delete m; }

void fem_mesh_counts(const fem_mesh *m, int *V, int *F, int *H) {
  if (V) *V = (int)m->mesh.n_vertices();
  if (F) *F = (int)m->mesh.n_faces();
  if (H) *H = m->mesh.H;
}

void fem_mesh_arrays(const fem_mesh *m, const double **points, const fem_uint **tri) {
  if (points) *points = m->mesh.points.data();
  if (tri) *tri = m->mesh.faces.data();
}

void fem_mesh_topology(const fem_mesh *m, const int32_t **opp, const fem_uint **he_from,
                       const fem_uint **he_to, const int32_t **rowptr, const fem_uint **col,
                       const int32_t **row_he) {
  if (opp) *opp = m->mesh.opp.data();
  if (he_from) *he_from = m->mesh.he_from.data();
  if (he_to) *he_to = m->mesh.he_to.data();
  if (rowptr) *rowptr = m->mesh.rowptr.data();
  if (col) *col = m->mesh.col.data();
  if (row_he) *row_he = m->mesh.row_he.data();
}

// ---- partition -----------------------------------------------------------------

int fem_partition_vertices(const fem_mesh *m, int n, int64_t *npart) {
  return fem::guarded("fem_partition_vertices", [&]() -> int {
  if (!m || !npart) return fail(FEM_ERR_INVALID, "fem_partition_vertices: null");
  int rc = fem::partitionVertices(m->mesh, n, npart);
  return rc == FEM_OK ? rc : fail(rc, "fem_partition_vertices: need 1 <= n <= V");
  });
}

int fem_partition_vertices_graph(const fem_mesh *m, int n, int64_t *npart) {
  return fem::guarded("fem_partition_vertices_graph", [&]() -> int {
  if (!m || !npart) return fail(FEM_ERR_INVALID, "fem_partition_vertices_graph: null");
  int rc = fem::partitionVerticesGraph(m->mesh, n, npart);
  return rc == FEM_OK ? rc : fail(rc, "fem_partition_vertices_graph: need 1 <= n <= V and a mesh with topology");
  });
}

int fem_partition_vertices_auto(const fem_mesh *m, int n, int64_t *npart, int *chosen) {
  return fem::guarded("fem_partition_vertices_auto", [&]() -> int {
  if (!m || !npart) return fail(FEM_ERR_INVALID, "fem_partition_vertices_auto: null");
  int rc = fem::partitionVerticesAuto(m->mesh, n, npart, chosen);
  return rc == FEM_OK ? rc : fail(rc, "fem_partition_vertices_auto: need 1 <= n <= V");
  });
}

int fem_partition_vertices_multilevel(const fem_mesh *m, int n, int64_t *npart, int64_t *epart, int64_t *dual_cut) {
  return fem::guarded("fem_partition_vertices_multilevel", [&]() -> int {
  if (!m || !npart) return fail(FEM_ERR_INVALID, "fem_partition_vertices_multilevel: null");
  std::vector<int64_t> ep;
  if (!epart && dual_cut) { ep.resize(m->mesh.n_faces()); epart = ep.data(); }
  int rc = fem::partitionVerticesMultilevel(m->mesh, n, npart, epart);
  if (rc != FEM_OK) return fail(rc, "fem_partition_vertices_multilevel: need 1 <= n <= F and a mesh with topology");
  if (dual_cut) *dual_cut = fem::dualEdgeCut(m->mesh, epart);
  return FEM_OK;
  });
}

int fem_mesh_wide_rows_first(const fem_mesh *m, int width, int64_t *new_index, fem_mesh **out) {
  return fem::guarded("fem_mesh_wide_rows_first", [&]() -> int {
  if (!m || !out || width < 1) return fail(FEM_ERR_INVALID, "fem_mesh_wide_rows_first: bad argument");
  *out = nullptr;
  const fem::SimpleTriMesh &om = m->mesh;
  const int64_t V = om.n_vertices(), F = om.n_faces();
  if (om.H == 0 && F > 0) return fail(FEM_ERR_INVALID, "fem_mesh_wide_rows_first: mesh without topology");
  // stable counting sort by the number of entries beyond `width`, most first
  auto extra = [&](int64_t v) { const int d = om.rowptr[v + 1] - om.rowptr[v]; return d > width ? d - width : 0; };
  int maxExtra = 0;
  int64_t wide = 0;
  for (int64_t v = 0; v < V; v++) { const int x = extra(v); maxExtra = std::max(maxExtra, x); wide += x > 0; }
  if (wide == 0) {
    if (new_index) for (int64_t v = 0; v < V; v++) new_index[v] = v;
    return 0;
  }
  std::vector<int64_t> start((size_t)maxExtra + 2, 0);
  for (int64_t v = 0; v < V; v++) start[(size_t)(maxExtra - extra(v)) + 1]++;
  for (int k = 0; k <= maxExtra; k++) start[(size_t)k + 1] += start[(size_t)k];
  std::vector<int64_t> pos((size_t)V);
  for (int64_t v = 0; v < V; v++) pos[(size_t)v] = start[(size_t)(maxExtra - extra(v))]++;
  fem_mesh *o = new (std::nothrow) fem_mesh();
  if (!o) return fail(FEM_ERR_ALLOC, "out of memory");
  o->mesh.points.resize(3 * (size_t)V);
  o->mesh.faces.resize(3 * (size_t)F);
  const bool col = om.colors.size() == om.points.size();
  if (col) o->mesh.colors.resize(3 * (size_t)V);
  for (int64_t v = 0; v < V; v++)
    for (int k = 0; k < 3; k++) {
      o->mesh.points[3 * (size_t)pos[(size_t)v] + k] = om.points[3 * (size_t)v + k];
      if (col) o->mesh.colors[3 * (size_t)pos[(size_t)v] + k] = om.colors[3 * (size_t)v + k];
    }
  for (size_t i = 0; i < 3 * (size_t)F; i++) o->mesh.faces[i] = (fem_uint)pos[om.faces[i]];
  const int rc = o->mesh.update_topology();
  if (rc != FEM_OK) { delete o; return fail(rc, "fem_mesh_wide_rows_first: renumbered mesh invalid"); }
  if (new_index) std::copy(pos.begin(), pos.end(), new_index);
  *out = o;
  return (int)std::min<int64_t>(wide, 0x7fffffff);
  });
}

int fem_partition_apply(const fem_mesh *original, int n, const int64_t *npart, int64_t *perm_v,
                        int64_t *perm_f, fem_uint *node_parts, fem_uint *element_parts,
                        fem_mesh **ordered) {
  return fem::guarded("fem_partition_apply", [&]() -> int {
  if (!original || !npart || !perm_v || !perm_f || !node_parts || !element_parts || n < 1)
    return fail(FEM_ERR_INVALID, "fem_partition_apply: bad argument");
  const fem::SimpleTriMesh &om = original->mesh;
  const int64_t V = om.n_vertices(), F = om.n_faces();
  for (int64_t i = 0; i < V; i++)
    if (npart[i] < 0 || npart[i] >= n) return fail(FEM_ERR_INVALID, "fem_partition_apply: part id out of range");
  int64_t cut = fem::edgeCut(om, npart);
  memcpy(perm_v, npart, sizeof(int64_t) * (size_t)V);
  fem::faceParts(om, npart, n, perm_f);
  fem::createOrderArray(node_parts, perm_v, V, n);
  fem::createOrderArray(element_parts, perm_f, F, n + 1);
  if (ordered) {
    fem_mesh *o = new (std::nothrow) fem_mesh();
    if (!o) return fail(FEM_ERR_ALLOC, "out of memory");
    fem::reorderMesh(om, &o->mesh, perm_v, perm_f);
    int rc = o->mesh.update_topology();
    if (rc != FEM_OK) { delete o; return fail(rc, "fem_partition_apply: reordered mesh invalid"); }
    *ordered = o;
  }
  return (int)(cut > 0x7fffffff ? 0x7fffffff : cut);
  });
}

int fem_partition(const fem_mesh *original, int n, int64_t *perm_v, int64_t *perm_f,
                  fem_uint *node_parts, fem_uint *element_parts, fem_mesh **ordered) {
  return fem::guarded("fem_partition", [&]() -> int {
  if (!original || !perm_v) return fail(FEM_ERR_INVALID, "fem_partition: null");
  std::vector<int64_t> npart(original->mesh.n_vertices());
  int rc = fem_partition_vertices(original, n, npart.data());
  if (rc != FEM_OK) return rc;
  return fem_partition_apply(original, n, npart.data(), perm_v, perm_f, node_parts, element_parts, ordered);
  });
}

// ---- assembly + CPU operators ------------------------------------------------------

int fem_mesh_init_stats(fem_mesh *m, int precision) {
  return fem::guarded("fem_mesh_init_stats", [&]() -> int {
  if (!m || (precision != FEM_PREC_MIXED && precision != FEM_PREC_F64))
    return fail(FEM_ERR_INVALID, "fem_mesh_init_stats: precision must be FEM_PREC_MIXED or FEM_PREC_F64");
  fem::initMeshStats(m->mesh, m->stats, precision);
  return FEM_OK;
  });
}

int fem_mesh_init_stats_shard(fem_mesh *m, int precision, const fem_shard *shard) {
  return fem::guarded("fem_mesh_init_stats_shard", [&]() -> int {
  if (!m || !shard || (precision != FEM_PREC_MIXED && precision != FEM_PREC_F64))
    return fail(FEM_ERR_INVALID, "fem_mesh_init_stats_shard: bad argument");
  for (fem::uint v : shard->s.loc2glob)
    if (v >= m->mesh.n_vertices()) return fail(FEM_ERR_INVALID, "fem_mesh_init_stats_shard: the shard is not a shard of this mesh");
  fem::initMeshStatsRows(m->mesh, m->stats, precision, shard->s.loc2glob);
  return FEM_OK;
  });
}

int fem_mesh_stats(const fem_mesh *m, const double **meshAngle, const double **meshCot,
                   const double **meshArea, const double **areaPerVtx, const double **wHej,
                   const double **wVtx, double *maxEdgeLen, const double **gradVec12,
                   const double **gradVec13) {
  return fem::guarded("fem_mesh_stats", [&]() -> int {
  if (!m || m->stats.precision < 0) return fail(FEM_ERR_INVALID, "fem_mesh_stats: stats not initialised");
  const fem::MeshStats &s = m->stats;
  if (meshAngle) *meshAngle = s.meshAngle.data();
  if (meshCot) *meshCot = s.meshCot.data();
  if (meshArea) *meshArea = s.meshArea.data();
  if (areaPerVtx) *areaPerVtx = s.areaPerVtx.data();
  if (wHej) *wHej = s.wHej.data();
  if (wVtx) *wVtx = s.wVtx.data();
  if (maxEdgeLen) *maxEdgeLen = s.maxEdgeLen;
  if (gradVec12) *gradVec12 = s.gradVec12.data();
  if (gradVec13) *gradVec13 = s.gradVec13.data();
  return FEM_OK;
  });
}

int fem_mesh_spectrum_bound(const fem_mesh *m, double *bound) {
  return fem::guarded("fem_mesh_spectrum_bound", [&]() -> int {
  if (!m || !bound || m->stats.precision < 0) return fail(FEM_ERR_INVALID, "fem_mesh_spectrum_bound: stats not initialised");
  *bound = fem::laplacianSpectrumBound(m->mesh, m->stats);
  return FEM_OK;
  });
}

int fem_compute_laplacian_cpu(const fem_mesh *m, const double *fn, double *lap) {
  return fem::guarded("fem_compute_laplacian_cpu", [&]() -> int {
  if (!m || m->stats.precision < 0 || !fn || !lap) return fail(FEM_ERR_INVALID, "fem_compute_laplacian_cpu: bad argument");
  fem::computeLaplacianCPU(m->mesh, fn, lap, m->stats.wHej.data(), m->stats.wVtx.data());
  return FEM_OK;
  });
}

int fem_compute_face_gradients_cpu(const fem_mesh *m, const double *fn, double *faceGrad) {
  return fem::guarded("fem_compute_face_gradients_cpu", [&]() -> int {
  if (!m || m->stats.precision < 0 || !fn || !faceGrad) return fail(FEM_ERR_INVALID, "fem_compute_face_gradients_cpu: bad argument");
  fem::computeFaceGradientsCPU(m->mesh, m->stats.gradVec12.data(), m->stats.gradVec13.data(), faceGrad, fn, m->stats.precision);
  return FEM_OK;
  });
}

int fem_compute_vertex_gradients_cpu(const fem_mesh *m, const double *faceGrad, double *vtxGrad) {
  return fem::guarded("fem_compute_vertex_gradients_cpu", [&]() -> int {
  if (!m || m->stats.precision < 0 || !faceGrad || !vtxGrad) return fail(FEM_ERR_INVALID, "fem_compute_vertex_gradients_cpu: bad argument");
  fem::computeVertexGradientsCPU(m->mesh, faceGrad, vtxGrad, m->stats.meshAngle.data(), m->stats.precision);
  return FEM_OK;
  });
}

// ---- reference layout ------------------------------------------------------------------

int fem_ref_layout_create(const fem_mesh *ordered, int n, const fem_uint *node_parts,
                          const fem_uint *element_parts, fem_ref_layout **out) {
  return fem::guarded("fem_ref_layout_create", [&]() -> int {
  if (!ordered || !node_parts || !element_parts || !out) return fail(FEM_ERR_INVALID, "fem_ref_layout_create: null");
  fem_ref_layout *l = new (std::nothrow) fem_ref_layout();
  if (!l) return fail(FEM_ERR_ALLOC, "out of memory");
  int rc = fem::packRefLayout(ordered->mesh, ordered->stats, n, node_parts, element_parts, l->l);
  if (rc != FEM_OK) { delete l; return fail(rc, "fem_ref_layout_create: stats missing or part arrays inconsistent"); }
  *out = l;
  return FEM_OK;
  });
}

void fem_ref_layout_destroy(fem_ref_layout *l) { delete l; }

void fem_ref_layout_sizes(const fem_ref_layout *l, fem_uint *max_vertex_count, fem_uint *vv_max_neighbors,
                          fem_uint *max_face_count, fem_uint *n_halo_faces) {
  if (max_vertex_count) *max_vertex_count = l->l.max_vertex_count;
  if (vv_max_neighbors) *vv_max_neighbors = l->l.vv_max_neighbors;
  if (max_face_count) *max_face_count = l->l.max_face_count;
  if (n_halo_faces) *n_halo_faces = (fem_uint)l->l.halo_faces.size();
}

void fem_ref_layout_arrays(const fem_ref_layout *l, const fem_uint **halo_faces_keys,
                           const fem_uint **halo_faces, const fem_uint **block_face_count,
                           const double **vertex_weights, const fem_uint **nbr_v,
                           const double **vv_weights, const fem_uint **face_vertices,
                           const double **fv_weights_new, const fem_float4 **he_grads) {
  if (halo_faces_keys) *halo_faces_keys = l->l.halo_faces_keys.data();
  if (halo_faces) *halo_faces = l->l.halo_faces.data();
  if (block_face_count) *block_face_count = l->l.block_face_count.data();
  if (vertex_weights) *vertex_weights = l->l.vertex_weights.data();
  if (nbr_v) *nbr_v = l->l.nbr_v.data();
  if (vv_weights) *vv_weights = l->l.vv_weights.data();
  if (face_vertices) *face_vertices = l->l.face_vertices.data();
  if (fv_weights_new) *fv_weights_new = l->l.fv_weights_new.data();
  if (he_grads) *he_grads = l->l.he_grads.data();
}

// ---- shards ---------------------------------------------------------------------------------

int fem_shard_create(const fem_mesh *ordered, int n, const fem_uint *node_parts, int rank, int nranks,
                     int krings, fem_shard **out) {
  return fem::guarded("fem_shard_create", [&]() -> int {
  if (!ordered || !node_parts || !out) return fail(FEM_ERR_INVALID, "fem_shard_create: null");
  fem_shard *s = new (std::nothrow) fem_shard();
  if (!s) return fail(FEM_ERR_ALLOC, "out of memory");
  int rc = fem::buildShard(ordered->mesh, n, node_parts, rank, nranks, krings, s->s);
  if (rc != FEM_OK) { delete s; return fail(rc, "fem_shard_create: need 1 <= nranks <= n, 0 <= rank < nranks, krings >= 1"); }
  *out = s;
  return FEM_OK;
  });
}

void fem_shard_destroy(fem_shard *s) { delete s; }

void fem_shard_info(const fem_shard *s, fem_uint *own_begin, fem_uint *own_end, fem_uint *n_local, fem_uint *n_send) {
  if (own_begin) *own_begin = s->s.own_begin;
  if (own_end) *own_end = s->s.own_end;
  if (n_local) *n_local = (fem_uint)s->s.loc2glob.size();
  if (n_send) *n_send = (fem_uint)s->s.send_idx.size();
}

void fem_shard_ranks(const fem_shard *s, int *rank, int *nranks, int *krings) {
  if (rank) *rank = s->s.rank;
  if (nranks) *nranks = s->s.nranks;
  if (krings) *krings = s->s.krings;
}

void fem_shard_arrays(const fem_shard *s, const fem_uint **loc2glob, const fem_uint **send_idx,
                      const int **send_counts, const int **recv_counts) {
  if (loc2glob) *loc2glob = s->s.loc2glob.data();
  if (send_idx) *send_idx = s->s.send_idx.data();
  if (send_counts) *send_counts = s->s.send_counts.data();
  if (recv_counts) *recv_counts = s->s.recv_counts.data();
}

}  // extern "C"
