/*
 * fem_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, serial restatement of the hot path of Blue-Robot/FEM (operator
 * assembly -> partition reorder contract -> device-layout packing -> explicit
 * Euler step), on flat arrays.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it, and only as the checker.  Nothing
 * under fem_amd/ links, imports or calls it.
 *
 * PARITY UNPINNED at bit level: the reference ships no tests, no golden
 * vectors and no build file, and cannot be compiled here (OpenMesh, METIS,
 * CUDA samples, GLEW/freeglut/FreeImage are absent).  This
 * restatement is pinned instead by (i) closed-form known answers
 * (tests/test_oracle_kat.py), (ii) an independent numpy/scipy formulation
 * (tests/test_oracle_scipy.py) and (iii) the only output artefact the
 * reference holds, ResultantPattern.off, as a qualitative fixture
 * (tests/test_resultant_pattern.py).
 *
 * Third-party behaviour that is NOT under /root/reference and is therefore
 * fixed here by convention (affects labelling / summation order only, i.e.
 * round-off, never the mathematical result):
 *   - OpenMesh half-edge numbering: here interior half-edge h = 3f+k runs
 *     tri[3f+k] -> tri[3f+(k+1)%3]; boundary half-edges are numbered 3F+b in
 *     ascending order of their interior opposite.
 *   - OpenMesh halfedge_handle(face): the LAST one, c->a for a face (a,b,c),
 *     so v1 = c, v2 = a, v3 = b in computeAreas/computeGradVectors/
 *     computeFaceGradientsCPU, and fh_iter/fv_iter yield a, b, c.
 *   - voh_iter order: ascending neighbour (to-vertex) index.
 *   - METIS_PartMeshDual: not restated; npart[] is an input.
 *
 * Two precision variants of everything the reference computes in a float/
 * double mix:  *_mixed = the reference's mix (Vec3f geometry and gradient
 * branch, double state/Laplacian),  *_f64 = all double (tolerance anchor).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FO_OK 0
#define FO_ERR_NONMANIFOLD -1
#define FO_ERR_ALLOC -2
#define FO_ERR_INPUT -3

/* ------------------------------------------------------------------ */
/* Topology: the part of OpenMesh the hot path leans on.               */
/* ------------------------------------------------------------------ */

typedef struct { uint64_t key; int32_t he; } fo_edge_rec;

static int fo_edge_cmp(const void *a, const void *b) {
  uint64_t ka = ((const fo_edge_rec *)a)->key, kb = ((const fo_edge_rec *)b)->key;
  return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

static int32_t fo_find(const fo_edge_rec *r, int n, uint64_t key) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    if (r[mid].key == key) return r[mid].he;
    if (r[mid].key < key) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

/*
 * Half-edge tables and the per-vertex outgoing-half-edge rows (the CSR
 * sparsity pattern: row i = 1-ring of i in ascending order).
 * Arrays sized for H <= 6F:  opp, he_from, he_to, col, row_he.  rowptr[V+1].
 * Replaces what OpenMesh builds inside read_mesh/add_face
 * (reference src/main.cpp:156-167, src/partitioning_common.cpp:150-163).
 */
int fo_topology(int V, int F, const uint32_t *tri, int32_t *opp,
                uint32_t *he_from, uint32_t *he_to, int32_t *rowptr,
                uint32_t *col, int32_t *row_he, int *H_out) {
  int H3 = 3 * F, h, b = 0;
  fo_edge_rec *rec = (fo_edge_rec *)malloc(sizeof(fo_edge_rec) * (size_t)(H3 > 0 ? H3 : 1));
  if (!rec) return FO_ERR_ALLOC;
  for (h = 0; h < H3; h++) {
    int f = h / 3, k = h % 3;
    uint32_t a = tri[3 * f + k], c = tri[3 * f + (k + 1) % 3];
    if (a >= (uint32_t)V || c >= (uint32_t)V || a == c) { free(rec); return FO_ERR_INPUT; }
    he_from[h] = a; he_to[h] = c;
    rec[h].key = ((uint64_t)a << 32) | c; rec[h].he = h;
  }
  qsort(rec, (size_t)H3, sizeof(fo_edge_rec), fo_edge_cmp);
  for (h = 1; h < H3; h++)
    if (rec[h].key == rec[h - 1].key) { free(rec); return FO_ERR_NONMANIFOLD; }
  for (h = 0; h < H3; h++)
    opp[h] = fo_find(rec, H3, ((uint64_t)he_to[h] << 32) | he_from[h]);
  for (h = 0; h < H3; h++)
    if (opp[h] < 0) {                 /* boundary half-edge 3F+b: to(h) -> from(h) */
      int hb = H3 + b++;
      he_from[hb] = he_to[h]; he_to[hb] = he_from[h];
      opp[hb] = h; opp[h] = hb;
    }
  free(rec);
  {
    int H = H3 + b, i;
    fo_edge_rec *all = (fo_edge_rec *)malloc(sizeof(fo_edge_rec) * (size_t)(H > 0 ? H : 1));
    if (!all) return FO_ERR_ALLOC;
    for (h = 0; h < H; h++) { all[h].key = ((uint64_t)he_from[h] << 32) | he_to[h]; all[h].he = h; }
    qsort(all, (size_t)H, sizeof(fo_edge_rec), fo_edge_cmp);
    for (i = 0; i <= V; i++) rowptr[i] = 0;
    for (h = 0; h < H; h++) rowptr[(all[h].key >> 32) + 1]++;
    for (i = 0; i < V; i++) rowptr[i + 1] += rowptr[i];
    for (h = 0; h < H; h++) { col[h] = (uint32_t)(all[h].key & 0xffffffffu); row_he[h] = all[h].he; }
    free(all);
    *H_out = H;
  }
  return FO_OK;
}

/* next/prev of an interior half-edge inside its face */
static inline int fo_next(int h) { return (h / 3) * 3 + (h % 3 + 1) % 3; }
static inline int fo_prev(int h) { return (h / 3) * 3 + (h % 3 + 2) % 3; }

/* A3  computeLaplacianWeights, reference src/mesh_function.cpp:219-271.
 * wHej[h: i->j] = (cot[next h] [h interior] + cot[next opp h] [opp interior])
 *                 / (2 areaPerVtx[i]);  wVtx[i] -= wHej[h]  (wVtx zeroed at
 * src/mesh_function.cpp:49-52).  All double in the reference. */
static void fo_compute_laplacian_weights(int V, int F, int H, const int32_t *opp,
                                         const uint32_t *he_from, const double *meshCot,
                                         const double *areaPerVtx, double *wHej, double *wVtx) {
  int h;
  for (h = 0; h < V; h++) wVtx[h] = 0;
  for (h = 0; h < H; h++) {
    uint32_t v1 = he_from[h];
    int ho = opp[h];
    double wg = 0;
    if (h < 3 * F) wg += meshCot[fo_next(h)];
    if (ho < 3 * F) wg += meshCot[fo_next(ho)];
    wHej[h] = wg / (2 * areaPerVtx[v1]);
    wVtx[v1] -= wHej[h];
  }
}

/* C1  computeLaplacianCPU, reference src/mesh_function.h:6-33.
 * lap[i] = fn[i] wVtx[i] + sum over outgoing half-edges fn[to] wHej[h]. */
void fo_laplacian(int V, const int32_t *rowptr, const uint32_t *col, const int32_t *row_he,
                  const double *fn, double *lap, const double *wHej, const double *wVtx) {
  int i;
#pragma omp parallel for schedule(static)
  for (i = 0; i < V; i++) {
    int e;
    lap[i] = fn[i] * wVtx[i];
    for (e = rowptr[i]; e < rowptr[i + 1]; e++) lap[i] += fn[col[e]] * wHej[row_he[e]];
  }
}

/* threads used by the step loops (1 unless built with -fopenmp); for the cpu_baseline leg */
#ifdef _OPENMP
#include <omp.h>
int fo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int fo_set_threads(int n) { (void)n; return 1; }
#endif

/* ------------------------------------------------------------------ */
/* Assembly + step, generated twice (geometry type float / double).    */
/* ------------------------------------------------------------------ */

#define GT float
#define GSQRT sqrtf
#define GACOS acosf
#define FN(x) x##_mixed
#include "fem_oracle_impl.inc"
#undef GT
#undef GSQRT
#undef GACOS
#undef FN

#define GT double
#define GSQRT sqrt
#define GACOS acos
#define FN(x) x##_f64
#include "fem_oracle_impl.inc"
#undef GT
#undef GSQRT
#undef GACOS
#undef FN

/* ------------------------------------------------------------------ */
/* Initial condition (reference src/main.cpp:178-186).                 */
/* rand() is replaced by an explicit 31-bit Lehmer generator           */
/* (x <- 48271 x mod 2^31-1): the reference's own stream is unpinned   */
/* (unseeded libc rand(), re-seeded by METIS).  betaMax is the float   */
/* literal 0.01f widened to double, as at src/main.cpp:50.             */
/* ------------------------------------------------------------------ */
void fo_initial_condition(int V, uint32_t seed, double nMax, double *nFn, double *cFn) {
  const double betaMax = (double)0.01f;
  uint64_t x = seed % 2147483647u;
  int i;
  if (x == 0) x = 1;
  for (i = 0; i < V; i++) {
    double beta;
    x = (x * 48271u) % 2147483647u;
    beta = -0.5 * betaMax + (double)x * betaMax / 2147483647.0;
    nFn[i] = nMax - beta;
    cFn[i] = 1 / (1 + nFn[i]);
  }
}

/* ------------------------------------------------------------------ */
/* Partition contract after METIS                                       */
/* (reference src/partitioning_common.cpp:78-176).                      */
/* ------------------------------------------------------------------ */

/* R1: face part = common part of its 3 vertices, else n ("cut" bucket).
 * partitioning_common.cpp:78-95 */
void fo_face_parts(int F, const uint32_t *tri, const int64_t *npart, int n, int64_t *epart) {
  int i;
  for (i = 0; i < F; i++) {
    int64_t p1 = npart[tri[3 * i]], p2 = npart[tri[3 * i + 1]], p3 = npart[tri[3 * i + 2]];
    epart[i] = (p1 == p2 && p2 == p3) ? p1 : n;
  }
}

/* R2: stable counting sort; part[] is overwritten with the new position of
 * each old index, parts[0..n] receives the start offsets.
 * partitioning_common.cpp:112-139 */
int fo_create_order_array(uint32_t *parts, int64_t *part, int N, int n) {
  int64_t *freq = (int64_t *)calloc((size_t)n, sizeof(int64_t));
  int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
  int64_t i;
  if (!freq || !tmp) { free(freq); free(tmp); return FO_ERR_ALLOC; }
  for (i = 0; i < N; i++) {
    if (part[i] < 0 || part[i] >= n) { free(freq); free(tmp); return FO_ERR_INPUT; }
    tmp[i] = freq[part[i]]++;
  }
  for (i = 1; i < n; i++) freq[i] += freq[i - 1];
  for (i = 0; i < N; i++) {
    int64_t p = part[i];
    part[i] = tmp[i] + (p == 0 ? 0 : freq[p - 1]);
  }
  for (i = 0; i < n + 1; i++) parts[i] = (uint32_t)(i == 0 ? 0 : freq[i - 1]);
  free(freq); free(tmp);
  return FO_OK;
}

/* R3: rebuild the mesh in the new order; corner order of each face is kept.
 * perm_v/perm_f = new position of old index (the output of R2).
 * partitioning_common.cpp:141-176 */
void fo_reorder_mesh(int V, int F, const double *pts, const uint32_t *tri,
                     const int64_t *perm_v, const int64_t *perm_f,
                     double *pts_out, uint32_t *tri_out) {
  int i, k;
  for (i = 0; i < V; i++)
    for (k = 0; k < 3; k++) pts_out[3 * perm_v[i] + k] = pts[3 * i + k];
  for (i = 0; i < F; i++)
    for (k = 0; k < 3; k++) tri_out[3 * perm_f[i] + k] = (uint32_t)perm_v[tri[3 * i + k]];
}

/* ------------------------------------------------------------------ */
/* Device-layout packing (reference src/main.cpp:207-342, App. A).      */
/* ------------------------------------------------------------------ */

static int fo_u32_cmp(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* Halo faces: per partition the ascending set of faces incident to its
 * vertices whose index lies outside [element_parts[p], element_parts[p+1]).
 * halo_faces sized 3F.  main.cpp:209-232 */
int fo_halo_faces(int V, int F, const uint32_t *tri, int n, const uint32_t *node_parts,
                  const uint32_t *element_parts, uint32_t *halo_keys, uint32_t *halo_faces) {
  /* vertex -> incident faces (CSR) */
  int32_t *vf_ptr = (int32_t *)calloc((size_t)V + 2, sizeof(int32_t));
  uint32_t *vf = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(3 * F > 0 ? 3 * F : 1));
  int i, p, total = 0;
  if (!vf_ptr || !vf) { free(vf_ptr); free(vf); return FO_ERR_ALLOC; }
  for (i = 0; i < 3 * F; i++) vf_ptr[tri[i] + 2]++;
  for (i = 0; i < V; i++) vf_ptr[i + 2] += vf_ptr[i + 1];
  for (i = 0; i < 3 * F; i++) vf[vf_ptr[tri[i] + 1]++] = (uint32_t)(i / 3);
  halo_keys[0] = 0;
  for (p = 0; p < n; p++) {
    int start = total, w, j;
    for (j = (int)node_parts[p]; j < (int)node_parts[p + 1]; j++) {
      int e;
      for (e = vf_ptr[j]; e < vf_ptr[j + 1]; e++) {
        uint32_t f = vf[e];
        if (f < element_parts[p] || f >= element_parts[p + 1]) halo_faces[total++] = f;
      }
    }
    qsort(halo_faces + start, (size_t)(total - start), sizeof(uint32_t), fo_u32_cmp);
    w = start;                                   /* std::set semantics: unique */
    for (j = start; j < total; j++)
      if (j == start || halo_faces[j] != halo_faces[j - 1]) halo_faces[w++] = halo_faces[j];
    total = w;
    halo_keys[p + 1] = (uint32_t)total;
  }
  free(vf_ptr); free(vf);
  return FO_OK;
}

/* sizes: max_vertex_count, vv_max_neighbors, max_face_count
 * main.cpp:246-250,261-270,290-294 */
void fo_pack_sizes(int V, int n, const uint32_t *node_parts, const uint32_t *element_parts,
                   const uint32_t *halo_keys, const int32_t *rowptr,
                   uint32_t *maxV, uint32_t *W, uint32_t *maxF) {
  int i;
  *maxV = 0; *W = 0; *maxF = 0;
  for (i = 0; i < n; i++) {
    uint32_t vc = node_parts[i + 1] - node_parts[i];
    uint32_t fc = element_parts[i + 1] - element_parts[i] + halo_keys[i + 1] - halo_keys[i];
    if (vc > *maxV) *maxV = vc;
    if (fc > *maxF) *maxF = fc;
  }
  for (i = 0; i < V; i++) {
    uint32_t d = (uint32_t)(rowptr[i + 1] - rowptr[i]);
    if (d > *W) *W = d;
  }
}

/* The padded SoA arrays `step` consumes (dense host form; index formulas in
 * SURVEY.md Appendix A).  grad12/grad13 are [3F] floats (the mixed variant's
 * gradVec12/13); he_grads is float4 as 4 floats.  main.cpp:236-341 */
void fo_pack_ref_layout(int V, int F, const uint32_t *tri, int n,
                        const uint32_t *node_parts, const uint32_t *element_parts,
                        const uint32_t *halo_keys, const uint32_t *halo_faces,
                        const int32_t *rowptr, const uint32_t *col, const int32_t *row_he,
                        const double *meshAngle, const double *wHej, const double *wVtx,
                        const float *grad12, const float *grad13,
                        uint32_t maxV, uint32_t W, uint32_t maxF,
                        uint32_t *block_face_count, double *vertex_weights,
                        uint32_t *nbr_v, double *vv_weights,
                        uint32_t *face_vertices, double *fv_weights, float *he_grads) {
  int i;
  (void)V; (void)F;
  memset(vertex_weights, 0, sizeof(double) * (size_t)n * maxV);
  memset(nbr_v, 0, sizeof(uint32_t) * (size_t)n * (W + 1) * maxV);
  memset(vv_weights, 0, sizeof(double) * (size_t)n * W * maxV);
  memset(face_vertices, 0, sizeof(uint32_t) * (size_t)n * 3 * maxF);
  memset(fv_weights, 0, sizeof(double) * (size_t)n * 3 * maxF);
  memset(he_grads, 0, sizeof(float) * 4 * (size_t)n * 2 * maxF);
  for (i = 0; i < n; i++) {
    uint32_t nint = element_parts[i + 1] - element_parts[i];
    uint32_t nhalo = halo_keys[i + 1] - halo_keys[i];
    uint32_t j, s;
    block_face_count[i] = nint + nhalo;
    for (j = 0; j < node_parts[i + 1] - node_parts[i]; j++) {
      uint32_t v = node_parts[i] + j, counter = 0;
      int e;
      vertex_weights[(size_t)i * maxV + j] = wVtx[v];
      for (e = rowptr[v]; e < rowptr[v + 1]; e++) {
        nbr_v[(size_t)i * (W + 1) * maxV + (size_t)(counter + 1) * maxV + j] = col[e];
        vv_weights[(size_t)i * W * maxV + (size_t)counter * maxV + j] = wHej[row_he[e]];
        counter++;
      }
      nbr_v[(size_t)i * (W + 1) * maxV + j] = counter;
    }
    for (s = 0; s < nint + nhalo; s++) {
      uint32_t f = s < nint ? element_parts[i] + s : halo_faces[halo_keys[i] + (s - nint)];
      int c, k;
      for (k = 0; k < 3; k++) {
        he_grads[((size_t)i * 2 * maxF + s) * 4 + k] = grad12[3 * f + k];
        he_grads[((size_t)i * 2 * maxF + maxF + s) * 4 + k] = grad13[3 * f + k];
      }
      /* fh_iter starts at halfedge_handle(f) = c->a, whose to-vertex is a:
       * corner c of the iteration has to-vertex tri[3f+c] and its half-edge
       * is the one ENDING there, i.e. prev of the half-edge leaving it. */
      for (c = 0; c < 3; c++) {
        int h_in = 3 * (int)f + (c + 2) % 3;         /* half-edge ending at tri[3f+c] */
        face_vertices[(size_t)i * 3 * maxF + (size_t)c * maxF + s] = tri[3 * f + c];
        fv_weights[(size_t)i * 3 * maxF + (size_t)c * maxF + s] = meshAngle[h_in];
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* A literal CPU model of stepKernel on the packed layout               */
/* (reference src/FEM_common.cu:82-166): used to check that the packed  */
/* arrays + kernel arithmetic agree with the mesh-level CPUrun above.   */
/* Serial, so the LDS "atomics" become ordered adds (face order).       */
/* ------------------------------------------------------------------ */
void fo_step_packed(int n, const uint32_t *node_parts, const uint32_t *block_face_count,
                    uint32_t maxV, uint32_t W, uint32_t maxF,
                    const double *vertex_weights, const uint32_t *nbr_v, const double *vv_weights,
                    const uint32_t *face_vertices, const double *fv_weights, const float *he_grads,
                    const double *model, double dt, const double *fn_src, double *fn_dst) {
  const double nMax = model[0], D = model[1], r = model[2], alpha = model[3], S = model[4];
  float *sn = (float *)malloc(sizeof(float) * 3 * (size_t)(maxV ? maxV : 1));
  float *sc = (float *)malloc(sizeof(float) * 3 * (size_t)(maxV ? maxV : 1));
  double *swg = (double *)malloc(sizeof(double) * (size_t)(maxV ? maxV : 1));
  int p;
  for (p = 0; p < n; p++) {
    uint32_t size = node_parts[p + 1] - node_parts[p], t;
    int k;
    for (t = 0; t < size; t++) { for (k = 0; k < 3; k++) sn[3 * t + k] = sc[3 * t + k] = 0.0f; swg[t] = 0.0; }
    for (t = 0; t < block_face_count[p]; t++) {
      int64_t idx[3];
      float g12[3], g13[3], nv[3], cv[3];
      double v1x, v1y, v12x, v12y, v13x, v13y;
      int j;
      for (j = 0; j < 3; j++) idx[j] = face_vertices[(size_t)p * 3 * maxF + (size_t)j * maxF + t];
      v1x = fn_src[2 * idx[2]]; v1y = fn_src[2 * idx[2] + 1];
      v12x = fn_src[2 * idx[0]] - v1x; v12y = fn_src[2 * idx[0] + 1] - v1y;
      v13x = fn_src[2 * idx[1]] - v1x; v13y = fn_src[2 * idx[1] + 1] - v1y;
      for (k = 0; k < 3; k++) {
        g12[k] = he_grads[((size_t)p * 2 * maxF + t) * 4 + k];
        g13[k] = he_grads[((size_t)p * 2 * maxF + maxF + t) * 4 + k];
        nv[k] = g12[k] * (float)v12x + g13[k] * (float)v13x;
        cv[k] = g12[k] * (float)v12y + g13[k] * (float)v13y;
      }
      for (j = 0; j < 3; j++) {
        int64_t li = idx[j] - (int64_t)node_parts[p];
        if (li >= 0 && li < (int64_t)size) {
          double weight = fv_weights[(size_t)p * 3 * maxF + (size_t)j * maxF + t];
          for (k = 0; k < 3; k++) {
            sn[3 * li + k] += (float)(nv[k] * weight);
            sc[3 * li + k] += (float)(cv[k] * weight);
          }
          swg[li] += weight;
        }
      }
    }
    for (t = 0; t < size; t++) {
      uint32_t i = node_parts[p] + t, end, j;
      double vW = vertex_weights[(size_t)p * maxV + t];
      double sx = fn_src[2 * i], sy = fn_src[2 * i + 1];
      double lx = sx * vW, ly = sy * vW, dotP, dauN, dauC, nx, ny;
      end = nbr_v[(size_t)p * (W + 1) * maxV + t];
      for (j = 0; j < end; j++) {
        uint32_t nIdx = nbr_v[(size_t)p * (W + 1) * maxV + (size_t)(j + 1) * maxV + t];
        double hW = vv_weights[(size_t)p * W * maxV + (size_t)j * maxV + t];
        lx += fn_src[2 * nIdx] * hW; ly += fn_src[2 * nIdx + 1] * hW;
      }
      if (swg[t] <= 0) dotP = 0;
      else {
        float w = (float)swg[t], d = 0.0f;
        for (k = 0; k < 3; k++) d += (sn[3 * t + k] / w) * (sc[3 * t + k] / w);
        dotP = d;
      }
      dauN = D * lx - alpha * sx * ly - alpha * dotP + S * r * sx * (nMax - sx);
      dauC = ly + S * (sx / (1 + sx) - sy);
      nx = dt * dauN + sx; ny = dt * dauC + sy;
      fn_dst[2 * i] = nx > 0 ? nx : 0.0;
      fn_dst[2 * i + 1] = ny > 0 ? ny : 0.0;
    }
  }
  free(sn); free(sc); free(swg);
}

/* ------------------------------------------------------------------ */
/* Colour map of formatKernel (reference src/FEM_common.cu:183-194):    */
/* n^ = (n - min)/(max - min); RGBA = (n^, 1-n^, 1-n^, 1).              */
/* ------------------------------------------------------------------ */
void fo_format(int V, const double *fn /* (n,c) AoS */, double *rgba /* 4V */) {
  double mn = fn[0], mx = fn[0];
  int i;
  for (i = 1; i < V; i++) { if (fn[2 * i] < mn) mn = fn[2 * i]; if (fn[2 * i] > mx) mx = fn[2 * i]; }
  for (i = 0; i < V; i++) {
    double nn = (fn[2 * i] - mn) / (mx - mn);
    rgba[4 * i] = nn; rgba[4 * i + 1] = 1 - nn; rgba[4 * i + 2] = 1 - nn; rgba[4 * i + 3] = 1.0;
  }
}
