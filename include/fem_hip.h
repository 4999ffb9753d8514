/*
 * fem_hip.h -- C-ABI of the MI355X-native FEM hot path (libfem_hip.so).
 *
 * Plain C: no HIP/CUDA headers, no C++ types.  Every entry point cites the
 * interface of the reference (Blue-Robot/FEM, paths under its src/) that it
 * replaces.  Pointers named dev_* / documented "device" are HIP device
 * pointers; everything else is host memory.  All functions returning int
 * return FEM_OK (0) or a negative fem_status; none aborts.
 */
#ifndef FEM_HIP_H
#define FEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned int fem_uint;                       /* the reference's `uint` */

/* Layout-identical stand-ins for CUDA's double2 / float4 (FEM_common.h:3-25). */
typedef struct __attribute__((aligned(16))) { double x, y; } fem_double2;
typedef struct __attribute__((aligned(16))) { float x, y, z, w; } fem_float4;

/* Model constants.  The reference compiles them into the .cu
 * (FEM_common.cu:9-13); here they are data.  fem_default_model() gives that
 * set: nMax=1, D=0.25, r=59.4537909143081, alpha=269, S=1. */
typedef struct { double nMax, D, r, alpha, S; } fem_model;

typedef enum {
  FEM_OK = 0,
  FEM_ERR_INVALID = -1,      /* bad argument / inconsistent sizes          */
  FEM_ERR_LIMIT = -2,        /* partition exceeds a kernel limit           */
  FEM_ERR_DEVICE = -3,       /* HIP runtime error (see fem_error_detail)   */
  FEM_ERR_ALLOC = -4,
  FEM_ERR_IO = -5,
  FEM_ERR_NONMANIFOLD = -6   /* a directed edge occurs twice               */
} fem_status;

#define FEM_PREC_MIXED 0     /* the reference's float/double mix           */
#define FEM_PREC_F64 1       /* all double                                 */
#define FEM_PREC_F32 2       /* native fp32 state + operator (plans only)  */

void fem_default_model(fem_model *m);
const char *fem_error_string(int status);
/* digest of the device sources (kernels_*.hpp, device.hip) and code-shaping flags this library was built from: measured
 * per-launch figures recorded under profiles/ (HBM traffic from counter passes) are tied to it */
const char *fem_kernel_build(void);
/* status of the last failing call on this thread (the `void step` shim has no
 * other way to report one) and the HIP error text that came with it */
int fem_last_error(void);
const char *fem_error_detail(void);
void fem_free(void *p);      /* frees arrays returned through T** out-params */

/* ------------------------------------------------------------------ */
/* 1. The device step: the reference's own symbol.                      */
/* ------------------------------------------------------------------ */

/* Replaces `extern "C" void step(...)`, FEM_common.h:3-25 / FEM_common.cu:168-180.
 * Same name, argument order and meaning: device pointers laid out as
 * main.cpp:207-342 packs them, byte pitches as
 * cudaMallocPitch/cudaMalloc3D returned them, blocks = number of partitions.
 * `threads` is accepted and ignored (the kernel picks its own wave64 geometry);
 * `smem_size` = 32 B x the largest partition's vertex count, as the reference
 * caller computes it (main.cpp:369), is read only for that count.
 * Asynchronous on the default stream; constants =
 * fem_default_model().  Errors: fem_last_error() right after the call is this
 * call's status (a successful step() resets it to FEM_OK). */
void step(fem_double2 *fn_src, fem_double2 *fn_dst, fem_uint *fv, double *fv_weights,
          fem_uint fv_pitchInBytes, fem_uint fvw_pitchInBytes, fem_uint *nbr, double *vtxW,
          fem_uint vw_pitchInBytes, double *vertex_weights, fem_uint vv_pitchInBytes,
          fem_uint vws_pitchInBytes, fem_uint vv_size, fem_float4 *grads,
          fem_uint he_pitchInBytes, fem_uint *parts_n, fem_uint *block_face_count,
          fem_uint blocks, fem_uint threads, double dt, fem_uint smem_size);

/* The same launch with a status, explicit constants and stream (hipStream_t
 * as void*; NULL = default stream).  max_part_vertices = vertex count of the
 * largest partition (with vv_size and the face pitch it sizes the kernel's LDS
 * work area: 4 (1 + vv_size) B per vertex + 24 B per face slot, FEM_ERR_LIMIT
 * beyond 160 KiB). */
int fem_step(fem_double2 *fn_src, fem_double2 *fn_dst, fem_uint *fv, double *fv_weights,
             fem_uint fv_pitchInBytes, fem_uint fvw_pitchInBytes, fem_uint *nbr, double *vtxW,
             fem_uint vw_pitchInBytes, double *vertex_weights, fem_uint vv_pitchInBytes,
             fem_uint vws_pitchInBytes, fem_uint vv_size, fem_float4 *grads,
             fem_uint he_pitchInBytes, fem_uint *parts_n, fem_uint *block_face_count,
             fem_uint blocks, double dt, fem_uint max_part_vertices, const fem_model *model,
             void *stream);

/* For callers of step() that include no HIP header: the reference caller's
 * cudaDeviceSynchronize() + cudaGetLastError() after the warm-up steps
 * (main.cpp:487-490) and its result read-backs cudaMemcpy(..., DeviceToHost)
 * (main.cpp:499,525; blocking, default stream). */
int fem_device_synchronize(void);
int fem_device_download(void *host_dst, uint64_t dev_src, size_t bytes);

/* ------------------------------------------------------------------ */
/* 2. Host interfaces the path keeps.                                   */
/* ------------------------------------------------------------------ */

/* .off / COFF load; replaces loadMesh -> OpenMesh::IO::read_mesh, main.cpp:156-167.
 * rgb (3 bytes per vertex) is NULL when the file has no colours.  Arrays are
 * malloc'ed; release with fem_free. */
int fem_off_read(const char *path, int *V, int *F, double **points, fem_uint **tri,
                 unsigned char **rgb);
/* Stanford PLY load (the format scans ship in; the reference reaches it through the same
 * OpenMesh::IO::read_mesh call): ascii and binary_little_endian, x y z of any scalar type, optional
 * red/green/blue, polygons fan-triangulated; same outputs as fem_off_read. */
int fem_ply_read(const char *path, int *V, int *F, double **points, fem_uint **tri,
                 unsigned char **rgb);
/* COFF writer ("visualisation output" without GL; file shape of
 * ResultantPattern.off:1-4).  rgb may be NULL (plain OFF). */
int fem_off_write(const char *path, int V, int F, const double *points, const fem_uint *tri,
                  const unsigned char *rgb);

/* Synthetic inputs (no counterpart in the reference; BASELINE.md workloads):
 * class-I geodesic icosahedron of frequency nu (V = 10 nu^2 + 2, F = 20 nu^2)
 * scaled so the mean edge length is `edge`; and the nx x ny strip with the
 * connectivity and 45-degree tilt of ResultantPattern.off (nx=11, ny=41,
 * h=0.1 reproduces its geometry). */
int fem_make_icosphere(int nu, double edge, int *V, int *F, double **points, fem_uint **tri);
int fem_make_strip(int nx, int ny, double h, int *V, int *F, double **points, fem_uint **tri);
/* Upsampler for real scans (BASELINE config 4: "upsampled Stanford Lucy/bunny"): `levels` rounds of
 * 1-to-4 midpoint subdivision (V' = V + E, F' = 4F per round); old vertices keep index and
 * position, orientation and boundary are kept.  Outputs are fem_free'd by the caller. */
int fem_subdivide(int V, int F, const double *points, const fem_uint *tri, int levels, int *Vout, int *Fout,
                  double **points_out, fem_uint **tri_out);
/* "scanned-like" irregular closed surface for BASELINE config 4 (no scan data is available):
 * the icosphere with flip_fraction * E seeded random edge flips, high_valence_fraction * V vertices
 * driven to valence >= high_valence by fans of flips, and radial noise of noise_amp edge lengths */
int fem_make_scanned_like(int nu, double edge, fem_uint seed, double flip_fraction, double noise_amp,
                          double high_valence_fraction, int high_valence, int *V, int *F,
                          double **points, fem_uint **tri);

/* Initial condition of initializeData, main.cpp:178-186, with rand() replaced
 * by an explicit Lehmer generator (x <- 48271 x mod 2^31-1) seeded by `seed`. */
void fem_initial_condition(int V, fem_uint seed, double nMax, double *nFn, double *cFn);

/* Mesh handle = points + faces + the half-edge tables OpenMesh would hold. */
typedef struct fem_mesh fem_mesh;
int fem_mesh_create(int V, int F, const double *points, const fem_uint *tri, fem_mesh **out);
void fem_mesh_destroy(fem_mesh *m);
void fem_mesh_counts(const fem_mesh *m, int *V, int *F, int *H);
void fem_mesh_arrays(const fem_mesh *m, const double **points, const fem_uint **tri);
/* half-edge tables: opp/he_from/he_to [H]; CSR rows of outgoing half-edges
 * (ascending neighbour id): rowptr [V+1], col/row_he [H] */
void fem_mesh_topology(const fem_mesh *m, const int32_t **opp, const fem_uint **he_from,
                       const fem_uint **he_to, const int32_t **rowptr, const fem_uint **col,
                       const int32_t **row_he);

/* Replaces `extern "C" int partition(SimpleTriMesh, SimpleTriMesh*, uint**,
 * uint**, int)`, partitioning_common.h:3-9 / partitioning_common.cpp:21-30, in
 * flat-array form.  Outputs: perm_v[V], perm_f[F] (new position of each old
 * index), node_parts[n+1], element_parts[n+2] (last bucket = cut faces), and
 * the reordered mesh.  Returns the edge cut (>= 0) or a negative fem_status.
 * METIS is replaced by a deterministic recursive-coordinate-bisection k-way split. */
int fem_partition(const fem_mesh *original, int n, int64_t *perm_v, int64_t *perm_f,
                  fem_uint *node_parts, fem_uint *element_parts, fem_mesh **ordered);
/* The stage METIS_PartMeshDual occupied (partitioning_common.cpp:51-58):
 * vertex -> part id in [0,n). */
int fem_partition_vertices(const fem_mesh *m, int n, int64_t *npart);
/* Same stage, coordinate-free: recursive bisection of the vertex graph (breadth-first halves from
 * a pseudo-peripheral vertex + pairwise boundary refinement); same balance (part sizes differ by
 * at most one), same hierarchical numbering, deterministic.  For surfaces where a coordinate
 * split makes disconnected parts (folds, thin sheets); feed the result to fem_partition_apply. */
int fem_partition_vertices_graph(const fem_mesh *m, int n, int64_t *npart);
/* The two above under one call (the drivers' default, FEM_PARTITIONER=auto): coordinate bisection, kept unless its
 * edge cut is far above what compact parts have (cut edges / (parts x sqrt(rows per part)) > 6: the sign of a surface
 * whose coordinates mislead -- folds, rolled or stacked sheets -- where its parts fall apart into pieces and the fused
 * kernel's ring redundancy nearly doubles); then graph bisection, if that cuts fewer edges.  *chosen (optional):
 * 0 = coordinate, 1 = graph.  Both are recursive bisections, so the part numbering is hierarchical either way: a rank's
 * contiguous range of parts (fem_shard_create) is a subtree, i.e. the split across GPUs is the top of the same recursion
 * that makes the tile-sized parts, with part sizes differing by at most one row. */
int fem_partition_vertices_auto(const fem_mesh *m, int n, int64_t *npart, int *chosen);
/* Same stage the way METIS_PartMeshDual does it (partitioning_common.cpp:51-58, ncommon = 2): multilevel k-way
 * partition of the DUAL graph (elements adjacent across an edge) -- heavy-edge-matching coarsening, weighted
 * coordinate bisection of the coarsest graph, greedy k-way boundary refinement at every level, part weights within
 * 3 % of the mean (how the per-block caps of main.cpp:357-364 enter) -- then the node partition induced from it
 * (a vertex goes where most of its faces are).  Deterministic.  epart[F] (element partition) and dual_cut (dual
 * edges cut) are optional outputs.  Feed npart to fem_partition_apply.  METIS itself is absent and unpinned, so raw
 * part ids cannot be compared with the original binary's; the contract after this stage is bit-exact. */
int fem_partition_vertices_multilevel(const fem_mesh *m, int n, int64_t *npart, int64_t *epart, int64_t *dual_cut);
/* The deterministic stage after it (partitioning_common.cpp:78-176), given any
 * npart[]: same outputs as fem_partition. */
int fem_partition_apply(const fem_mesh *original, int n, const int64_t *npart, int64_t *perm_v,
                        int64_t *perm_f, fem_uint *node_parts, fem_uint *element_parts,
                        fem_mesh **ordered);

/* Numbering pre-pass for irregular meshes, to run BEFORE the partition stage.  createOrderArray keeps the input order
 * inside every part (partitioning_common.cpp:112-139: a stable counting sort), so the input numbering decides which
 * rows sit side by side in a part -- and the fused kernel's loop over the entries beyond `width` (6 in the register
 * tile) is run by a whole 64-row wave if any one of its rows has such an entry.  This renumbers the vertices stably by
 * the number of those entries, most first (faces keep their order and corner order): wide rows then lead every part
 * and fill a few waves instead of touching all of them (scanned-like 8.1M mesh: 228 -> 238 G DOF-it/s).  Returns the
 * number of wide rows (>= 0; 0: *out = NULL, keep the input) or a negative fem_status; new_index[V] (optional) = new
 * position of each old vertex.  No counterpart in the reference: its meshes are numbered by whoever wrote the file. */
int fem_mesh_wide_rows_first(const fem_mesh *m, int width, int64_t *new_index, fem_mesh **out);

/* Replaces initMeshStats(mesh, MeshStats&), mesh_function.cpp:27-67 (hot
 * fields of MeshStats, common.h:19-38).  precision = FEM_PREC_MIXED or
 * FEM_PREC_F64.  Arrays stay owned by the mesh handle. */
int fem_mesh_init_stats(fem_mesh *m, int precision);
int fem_mesh_stats(const fem_mesh *m, const double **meshAngle, const double **meshCot,
                   const double **meshArea, const double **areaPerVtx, const double **wHej,
                   const double **wVtx, double *maxEdgeLen, const double **gradVec12,
                   const double **gradVec13);

/* Upper bound on the largest eigenvalue of -L, L the assembled Laplacian (wHej, wVtx of the stats): the smaller of
 * Gershgorin's row bound and the largest per-triangle eigenvalue lambda_max(M_t^-1 K_t) (exact on regular meshes, where
 * Gershgorin is a third too large).  Both are rigorous, obtuse triangles included.  What the Chebyshev solver of the
 * semi-implicit step takes its coefficients from (fem_plan_run_imex_fused uses it by itself; sharded runs pass HALF of the
 * whole mesh's bound as `wsum_max`).  No counterpart in the reference (mesh_function.cpp:490-586 sketches implicit
 * stepping, with no solver). */
int fem_mesh_spectrum_bound(const fem_mesh *m, double *bound);

/* CPU operators kept by name (mesh_function.h:6-110), on the mesh handle. */
int fem_compute_laplacian_cpu(const fem_mesh *m, const double *fn, double *lap);
int fem_compute_face_gradients_cpu(const fem_mesh *m, const double *fn, double *faceGrad /*3F*/);
int fem_compute_vertex_gradients_cpu(const fem_mesh *m, const double *faceGrad, double *vtxGrad /*3V*/);

/* ------------------------------------------------------------------ */
/* 3. Reference device layout + harness (initializeGPUData / GPUrun).   */
/* ------------------------------------------------------------------ */

/* Replaces initializeGPUData, main.cpp:207-342: halo faces + padded SoA
 * arrays, on an ordered mesh whose stats are initialised. */
typedef struct fem_ref_layout fem_ref_layout;
int fem_ref_layout_create(const fem_mesh *ordered, int n, const fem_uint *node_parts,
                          const fem_uint *element_parts, fem_ref_layout **out);
void fem_ref_layout_destroy(fem_ref_layout *l);
void fem_ref_layout_sizes(const fem_ref_layout *l, fem_uint *max_vertex_count,
                          fem_uint *vv_max_neighbors, fem_uint *max_face_count,
                          fem_uint *n_halo_faces);
void fem_ref_layout_arrays(const fem_ref_layout *l, const fem_uint **halo_faces_keys,
                           const fem_uint **halo_faces, const fem_uint **block_face_count,
                           const double **vertex_weights, const fem_uint **nbr_v,
                           const double **vv_weights, const fem_uint **face_vertices,
                           const double **fv_weights_new, const fem_float4 **he_grads);

/* Replaces the allocation/upload/limit-check part of GPUrun, main.cpp:344-452:
 * pitched device copies of a layout plus two double2 state buffers. */
typedef struct fem_ref_device fem_ref_device;
int fem_ref_device_create(const fem_ref_layout *l, const fem_uint *node_parts, int V,
                          fem_ref_device **out);
void fem_ref_device_destroy(fem_ref_device *d);
int fem_ref_device_set_state(fem_ref_device *d, const fem_double2 *fn);   /* H2D, main.cpp:451 */
int fem_ref_device_get_state(fem_ref_device *d, fem_double2 *fn);         /* D2H, main.cpp:499  */
/* nsteps ping-pong `step` launches (main.cpp:558-562), no host sync inside. */
int fem_ref_device_run(fem_ref_device *d, int nsteps, double dt, const fem_model *model,
                       void *stream);
/* raw device pointers and pitches, for callers that invoke step() themselves;
 * args21 receives, in step()'s order, the 17 pointer/pitch/size arguments that
 * precede `blocks` as uint64 (pointers) */
void fem_ref_device_args(const fem_ref_device *d, uint64_t *fn_one, uint64_t *fn_two,
                         uint64_t *fv, uint64_t *fv_weights, fem_uint *fv_pitch,
                         fem_uint *fvw_pitch, uint64_t *nbr, uint64_t *vtxW, fem_uint *vw_pitch,
                         uint64_t *vertex_weights, fem_uint *vv_pitch, fem_uint *vws_pitch,
                         fem_uint *vv_size, uint64_t *grads, fem_uint *he_pitch,
                         uint64_t *parts_n, uint64_t *block_face_count, fem_uint *blocks,
                         fem_uint *threads, fem_uint *smem_size);

/* ------------------------------------------------------------------ */
/* 4. Native plan: the MI355X layout of the same operator.              */
/* ------------------------------------------------------------------ */

/* One plan = the assembled operator of an ordered mesh in the native device
 * layout (sliced ELL, 64-row slices, folded gradient coefficients)
 * plus two ping-pong state buffers.  Replaces initializeGPUData + GPUrun's
 * allocations for callers that do not need the reference's array shapes.
 * precision = FEM_PREC_F32 or FEM_PREC_F64.
 * State representation: the device buffers hold DEVIATIONS from the homogeneous
 * state (base_n, base_c) = (nMax, nMax/(1+nMax)) of the model given at creation
 * (NULL = fem_default_model), so that fp32 storage does not round at the
 * magnitude of n itself; set/get_state convert in double.  fem_plan_run may be
 * given different constants later; the rewrite stays exact for any of them. */
typedef struct fem_plan fem_plan;
typedef struct fem_shard fem_shard;   /* section 5 */
int fem_plan_create(const fem_mesh *ordered, int precision, const fem_model *model, fem_plan **out);
/* Same operator in the fused-tile layout: every workgroup stages one partition (+ ksteps rings
 * of neighbours) in LDS, keeps its operator rows in registers and takes up to `ksteps` time
 * steps per launch (fused multi-step kernel; 1 <= ksteps <= 8).  n/node_parts = the partition
 * ranges of the ordered mesh (partitioning_common.h:3-9 outputs); shard may be NULL.  Rows of more than 6
 * neighbours keep their first 6 entries in the register tile and the rest in LDS (up to 1536 such entries per
 * tile); partitions that do not fit a tile are split.  FEM_ERR_LIMIT only when a tile cannot be made to fit
 * (a vertex of valence > 1536).  fem_plan_run on such a plan issues ceil(nsteps / ksteps) launches. */
int fem_plan_create_tiled(const fem_mesh *ordered, int n, const fem_uint *node_parts,
                          const fem_shard *shard, int precision, const fem_model *model,
                          int ksteps, fem_plan **out);
/* General form of the three creators above.  flags: FEM_PLAN_DEVICE_ASSEMBLY keeps the mesh
 * connectivity, vertex positions and per-entry face tables on the device so that the operator can
 * be re-assembled there (fem_plan_assemble) -- initMeshStats on the GPU, mesh_function.cpp:27-67.
 * ksteps = 0: sliced-ELL layout; n/node_parts may then be 0/NULL. */
#define FEM_PLAN_DEVICE_ASSEMBLY 1u
#define FEM_PLAN_IMPLICIT 2u          /* sliced-ELL plan (whole mesh or a shard): enables fem_plan_run_imex[_sharded] (not with DEVICE_ASSEMBLY) */
/* fused fp64 plans (precision FEM_PREC_F64, ksteps >= 1, host assembly) in the reference's OWN precision mix: state,
 * differences, Laplacian weights and sums, and the update in fp64 (FN_TYPE = double, src/common.h:7-11; double2 fn and
 * fp64 ELL weights, FEM_common.cu:135-165), the gradient branch in fp32 (float4 grads, differences narrowed to float,
 * float3 accumulators: FEM_common.cu:99-132,153-158).  The operator stream is then 20 bytes per entry instead of 32
 * (fem_plan_operator_format = 2).  State buffers, set/get, halo packing: exactly as for any fp64 plan. */
#define FEM_PLAN_MIXED_RECORDS 4u
int fem_plan_create_ex(const fem_mesh *ordered, int n, const fem_uint *node_parts,
                       const fem_shard *shard, int precision, const fem_model *model, int ksteps,
                       unsigned flags, fem_plan **out);
/* vertex positions of the ordered mesh (3 doubles per vertex): new base positions + current positions */
int fem_plan_set_points(fem_plan *p, const double *points);
/* current positions = base positions scaled radially by 1 + amplitude * sin(q x + phase) cos(q y - phase/2)
 * sin(q z + phase/4) (q = wavenumber): the seeded smooth deformation of BASELINE config 5.  A shard's plan moves the
 * vertices of its own rows (owned + ghost) only: the deformation is a function of the base position, so ghost rows
 * deform locally and no exchange of positions is needed */
int fem_plan_deform(fem_plan *p, double amplitude, double wavenumber, double phase, void *stream);
/* Device assembly from the current positions, two kernels, async on `stream`:
 * per element: corner angles + cotangents (computeAngles, mesh_function.cpp:70-160), area
 * (computeAreas, :162-216), longest edge (computeMaxEdgeLength, :310-350), gradVec12/13
 * (computeGradVectors, :351-397); per operator row: lumped mass, cot-Laplacian weights
 * (computeLaplacianWeights, :219-271) and the folded vertex-gradient coefficients, written into the
 * plan's operator array.  Arithmetic type = the plan's precision (FEM_PREC_F32 ~ the reference's
 * Vec3f geometry). */
int fem_plan_assemble(fem_plan *p, void *stream);
/* One iteration of a deforming-mesh run (BASELINE config 5: re-assembly every step) in one kernel: the assembly of
 * fem_plan_assemble from the current positions, and one explicit time step (fem_plan_run(p, 1, ...)) taken with the
 * coefficients while they are still in the registers they were assembled in, so that the operator does not travel
 * through HBM between two kernels.  store_operator != 0 also writes the operator into the plan (for later
 * fem_plan_run calls on it); with 0 the plan's stored operator is left as it was.  maxEdgeLen is updated either way
 * (fem_plan_max_edge): dt is the caller's, e.g. from the previous iteration's edge length.  Fused for sliced-ELL
 * plans (ksteps = 0) whose rows are single fans of at most 12 entries; other plans run the two kernels in turn,
 * with the same result. */
int fem_plan_assemble_step(fem_plan *p, double dt, const fem_model *model, int store_operator, void *stream);
/* maxEdgeLen of the last fem_plan_assemble (dt = 0.1 maxEdgeLen^2, main.cpp:188); synchronises */
int fem_plan_max_edge(fem_plan *p, double *maxEdgeLen);
/* per-face records of the last assembly, 16 doubles per face:
 * {ang_a, ang_b, ang_c, area, cot_a, cot_b, cot_c, maxEdge, g12.x, g12.y, g12.z, 0, g13.x, g13.y, g13.z, 0} */
int fem_plan_face_records(fem_plan *p, double *rec16F);
/* the plan's operator array (4 values per entry, layout order), for comparing host- and device-assembled operators */
uint64_t fem_plan_coef_count(const fem_plan *p);
int fem_plan_download_coef(fem_plan *p, double *out);
/* Replaces `extern "C" void format(double2*, cudaGraphicsResource_t*, int)`, FEM_common.h:27-31 /
 * FEM_common.cu:181-230, without the GL interop: min/max of n over the first `nverts` vertices of the
 * current state (two device reductions, as thrust::min/max_element at :208-209), then
 * RGBA = (n^, 1-n^, 1-n^, 1), n^ = (n-min)/(max-min) (formatKernel, :183-194).  Outputs (host, any may
 * be NULL): rgba[4*nverts] doubles; rgb[3*nverts] 8-bit colours truncated as in ResultantPattern.off;
 * the extrema of n.  Synchronises.  Pass the result to fem_off_write for a COFF file. */
int fem_plan_format(fem_plan *p, fem_uint nverts, double *rgba, unsigned char *rgb, double *nmin,
                    double *nmax);
/* No counterpart in the reference.  For callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc: fem_plan_format allocates its partial-extrema
 * buffer, the extrema pair and a staging buffer on every call and synchronises; this variant runs the same three
 * kernels asynchronously on `stream` with the partials and extrema in a fixed array inside the library (16 400 B per
 * GPU, held until the process ends) and writes the colours straight into the caller's device memory:
 *   dev_rgba   device buffer of 4 * nverts doubles (32-byte aligned): RGBA as fem_plan_format's `rgba`;
 *   dev_minmax device buffer of 2 doubles (min, max of n), or 0.
 * It calls no hipMalloc/hipFree and no synchronisation.  nverts <= 67 108 864 (FEM_ERR_LIMIT beyond: the array
 * is sized for that).  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams,
 * two threads) race; calls ordered on one stream do not.  The first call on a device looks the array's address up
 * (make it outside a graph capture).  The GPU the plan was created on must be the current device (FEM_ERR_INVALID
 * otherwise: the kernels would run against another GPU's copy of the array), and `stream` a stream of that GPU. */
int fem_plan_format_noalloc(fem_plan *p, fem_uint nverts, uint64_t dev_rgba, uint64_t dev_minmax, void *stream);
/* geometry of a tiled plan: fused steps, tile count, largest tile (rows staged in LDS), operator
 * rows loaded per launch (sum over tiles), workgroup size and rows per thread; ksteps = 0 for a
 * sliced-ELL plan.  Any out-pointer may be NULL. */
int fem_plan_tile_info(const fem_plan *p, int *ksteps, fem_uint *ntiles, fem_uint *max_local,
                       uint64_t *rows_loaded, int *threads, int *rows_per_thread);
void fem_plan_destroy(fem_plan *p);
int fem_plan_set_state(fem_plan *p, const double *nFn, const double *cFn);
int fem_plan_get_state(fem_plan *p, double *nFn, double *cFn);
/* nsteps explicit Euler steps, ping-ponging the plan's two buffers; async on
 * `stream`.  One call = the reference's timed loop body (main.cpp:558-562). */
int fem_plan_run(fem_plan *p, int nsteps, double dt, const fem_model *model, void *stream);
/* Semi-implicit (IMEX Euler) steps -- no counterpart in the reference, which only hints at implicit
 * stepping (unused Jacobian scaffolding, mesh_function.cpp:490-586): reaction and
 * chemotaxis terms explicit, the diffusion terms implicit,
 *   (I - dt D L) n' = n + dt (-alpha n L c - alpha Gn.Gc + S r n (nMax - n)),
 *   (I - dt L)   c' = c + dt S (n/(1+n) - c),   then max(0, .),
 * each solved by `cg_iterations` Jacobi-preconditioned conjugate-gradient iterations on the
 * mass-weighted (symmetric positive definite) system, started from the right-hand side; all on the
 * device, no host synchronisation.  Plan must carry FEM_PLAN_IMPLICIT.  fem_plan_imex_residual:
 * preconditioned residual norm of the last solve relative to its first one (synchronises). */
int fem_plan_run_imex(fem_plan *p, int nsteps, double dt, const fem_model *model, int cg_iterations, void *stream);
/* The same semi-implicit steps on a FUSED-TILE plan (fem_plan_create_tiled, whole mesh; no flag needed), the two diffusion solves by Chebyshev iteration instead of conjugate gradients: its coefficients come
 * from a priori bounds on the spectrum (1 <= lambda <= 1 + 2 dt D max_v sum_j w_vj), so an iteration needs no dot
 * product -- it is to the tile kernel what an explicit step is, and ksteps iterations run per launch with the iterate in
 * LDS and the Laplacian weights in registers.  The iteration count follows from the bounds and rel_tol (the Chebyshev
 * error bound, in the energy norm relative to the first iterate's error; fem_plan_imex_fused_iterations returns it).
 * No counterpart in the reference (as fem_plan_run_imex).  4.1M vertices, dt = 30 x the explicit step: see DESIGN.md. */
int fem_plan_run_imex_fused(fem_plan *p, int nsteps, double dt, const fem_model *model, double rel_tol, void *stream);
int fem_plan_imex_fused_iterations(const fem_plan *p);
int fem_plan_imex_residual(fem_plan *p, double *rel_n, double *rel_c);
/* Residual-based stop (default 0 = every solve takes exactly cg_iterations): with rel_tol > 0 a solve ends as soon as
 * the preconditioned residual of BOTH fields is <= rel_tol times its first one; cg_iterations is then the upper bound.
 * The launch sequence stays fixed and free of host synchronisation: the kernels of a finished solve return at once
 * (a device flag), on every rank of a sharded solve alike (the sums are global).  fem_plan_imex_iterations: iterations
 * the last solve took (synchronises). */
int fem_plan_imex_tolerance(fem_plan *p, double rel_tol);
int fem_plan_imex_iterations(fem_plan *p, int *iterations);
/* The same steps on a shard's plan (fem_plan_create_ex(..., shard, ..., ksteps = 0, FEM_PLAN_IMPLICIT); one ghost ring is
 * enough): the conjugate-gradient iteration distributed over the ranks.  The library runs the kernels; the caller
 * supplies the two communication steps as callbacks, both asynchronous on `stream`:
 *   exchange(ctx, dev_vector, stream): fill the ghost rows [n_own, n_local) of a device vector of n_local interleaved
 *     pairs of the plan's scalar type from their owners (fem_plan_pack_vector gathers this rank's send rows);
 *   allreduce_sum(ctx, dev_doubles, count, stream): sum `count` doubles in device memory over all ranks, in place.
 * Per time step: one exchange of the right-hand side, then per CG iteration one exchange of the search direction and two
 * all-reduces of two doubles (p.q and r.z, owned rows only) -- the one place on this path where an all-reduce appears
 * (SURVEY.md 8f-4).  libfem_dist.so (fem_dist_run_imex) supplies both over RCCL. */
typedef struct {
  void *ctx;
  int (*exchange)(void *ctx, uint64_t dev_vector, void *stream);
  int (*allreduce_sum)(void *ctx, uint64_t dev_doubles, int count, void *stream);
} fem_imex_comm;
int fem_plan_run_imex_sharded(fem_plan *p, int nsteps, double dt, const fem_model *model, int cg_iterations, fem_uint n_own,
                              const fem_imex_comm *comm, void *stream);
/* fem_plan_run_imex_fused on a shard's fused-tile plan (fem_plan_create_tiled with the shard; the CALLER guarantees
 * ksteps <= the shard's ghost rings -- fem_dist_run_imex_fused checks it -- or ghost rows go stale inside a launch).
 * Only comm->exchange is used: the right-hand side once, iterate and direction after every launch of ksteps
 * iterations, the new state at the end -- no all-reduce, because the Chebyshev coefficients come from wsum_max, HALF an
 * upper bound on the spectrum of -L of the WHOLE mesh (fem_mesh_spectrum_bound / 2; max |wVtx| of MeshStats is the
 * Gershgorin value for meshes without obtuse angles), which the caller passes so that every rank runs the same sequence.  State ghosts must be current on entry and are on return. */
int fem_plan_run_imex_fused_sharded(fem_plan *p, int nsteps, double dt, const fem_model *model, double rel_tol, double wsum_max,
                                    const fem_imex_comm *comm, void *stream);
/* gathers this rank's send rows of ANY device vector of n_local pairs (fem_plan_pack_halo does it for the state) */
int fem_plan_pack_vector(fem_plan *p, uint64_t dev_vector, uint64_t dst_dev, void *stream);
/* device pointer of the current state buffer (interleaved (n - base_n,
 * c - base_c) pairs of the plan's scalar type), its vertex count, bytes per
 * scalar and the two base values */
int fem_plan_state(const fem_plan *p, uint64_t *dev_ptr, uint64_t *n_vertices, int *scalar_bytes,
                   double *base_n, double *base_c);
/* operator footprint in device memory (bytes read per step, excluding state) */
uint64_t fem_plan_operator_bytes(const fem_plan *p);
/* How the plan stores an operator entry (no counterpart in the reference, whose kernel reads its own layout):
 * 0 = four scalars of the plan's precision (w, Cx, Cy, Cz);
 * 1 = 12-byte records: w in fp32 and Cx, Cy, Cz as 21-bit signed integers against a per-row fp32 scale (rounding
 *     error at most 4.8e-7 of the row's largest gradient coefficient; the entries beyond the sixth of a wider row
 *     stay four floats).  Default for fp32 plans, fused or single-step, whose operator comes from the host assembly
 *     and that are not FEM_PLAN_IMPLICIT;
 *     FEM_TILE_PACKED=0 in the environment at plan creation keeps format 0.  fem_plan_download_coef returns the
 *     decoded values either way;
 * 2 = fused fp64 plans created with FEM_PLAN_MIXED_RECORDS: w in fp64, Cx, Cy, Cz in fp32 (20 bytes per entry). */
int fem_plan_operator_format(const fem_plan *p);

/* ------------------------------------------------------------------ */
/* 5. Sharding over the GPUs of one node (no counterpart in the         */
/*    reference, which is single-GPU).                    */
/* ------------------------------------------------------------------ */

/* Rank `rank` of `nranks` owns the contiguous partition range
 * [rank n / nranks, (rank+1) n / nranks) of an ordered mesh -- a contiguous
 * vertex range, as partitioning_common.cpp:132-136 lays partitions out -- plus
 * ghost copies of all vertices within `krings` edges of it.  Local numbering:
 * owned vertices (ascending), then ghosts (ascending global id, hence grouped
 * by owner rank).  With k rings a rank takes k steps between halo exchanges. */
int fem_shard_create(const fem_mesh *ordered, int n, const fem_uint *node_parts, int rank,
                     int nranks, int krings, fem_shard **out);
void fem_shard_destroy(fem_shard *s);
void fem_shard_info(const fem_shard *s, fem_uint *own_begin, fem_uint *own_end, fem_uint *n_local,
                    fem_uint *n_send);
void fem_shard_ranks(const fem_shard *s, int *rank, int *nranks, int *krings);
/* loc2glob[n_local]; send_idx[n_send] = local ids whose state goes out, grouped by
 * destination rank; send_counts/recv_counts[nranks] (all-to-all split sizes, in vertices) */
void fem_shard_arrays(const fem_shard *s, const fem_uint **loc2glob, const fem_uint **send_idx,
                      const int **send_counts, const int **recv_counts);
/* initMeshStats (mesh_function.cpp:27-67) for one rank: assembles only what the operator rows of the shard's local
 * vertices (owned + ghost rings) need -- the faces around them, the half-edges leaving them; all other entries of
 * the arrays fem_mesh_stats returns are 0, maxEdgeLen is the whole mesh's.  A listed row is bit-identical to the full
 * assembly's.  Plans of that shard can then be created without any rank assembling the whole mesh. */
int fem_mesh_init_stats_shard(fem_mesh *m, int precision, const fem_shard *shard);
/* Plan over a shard's local vertices; state arrays are n_local long, ghosts included. */
int fem_plan_create_shard(const fem_mesh *ordered, const fem_shard *shard, int precision,
                          const fem_model *model, fem_plan **out);
/* Gathers the current state of the shard's send vertices into dst_dev (n_send interleaved
 * pairs of the plan's scalar type, device memory), async on `stream`. */
int fem_plan_pack_halo(fem_plan *p, uint64_t dst_dev, void *stream);

/* Pieces of fem_plan_run for a caller that overlaps the halo exchange with the time steps (no
 * counterpart in the single-GPU reference; DESIGN.md "Multi-GPU").  The tiles of a shard's fused
 * plan are ordered [interior | boundary | ghost]: ghost = tiles of ghost rows; boundary = owned
 * tiles that read a ghost row or hold a row another rank needs; interior = the rest (a plan of
 * a whole mesh has interior tiles only).
 *   fem_plan_launch_tiles: 1 <= nsteps <= ksteps fused steps of tiles [first, first+count) from
 *     the current state buffer into the other ("next") one; does not flip;
 *   fem_plan_flip: next becomes current (call once every tile that matters has been launched);
 *   fem_plan_state_buffer / fem_plan_pack_halo_from: which = 0 current, 1 next. */
int fem_plan_tile_classes(const fem_plan *p, fem_uint *n_interior, fem_uint *n_boundary, fem_uint *n_ghost);
/* How fem_plan_run launches a fused plan (no counterpart in the reference, whose loop is one launch per step,
 * src/main.cpp:558-562): *groups = 0, one launch per group of ksteps steps; n > 0, *min_groups .. n groups per launch of the
 * persistent tile kernel (tiles handed from group to group through per-tile flags inside the launch; packed fp32 plans
 * of 192..2400 tiles by default, and only for launches of at least 12 groups; FEM_TILE_PERSIST=0 turns it off, =n forces
 * launches of 2..n groups at any size).  Results are bit-identical either way.
 * Should a tile wait beyond its bound for a neighbour (never observed), the launch ends early and the next
 * fem_plan_get_state returns FEM_ERR_DEVICE instead of a state. */
int fem_plan_persist_groups(const fem_plan *p, int *groups, int *min_groups);
int fem_plan_launch_tiles(fem_plan *p, fem_uint first, fem_uint count, int nsteps, double dt,
                          const fem_model *model, void *stream);
int fem_plan_flip(fem_plan *p);
int fem_plan_state_buffer(const fem_plan *p, int which, uint64_t *dev_ptr);
int fem_plan_pack_halo_from(fem_plan *p, int which, uint64_t dst_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FEM_HIP_H */
