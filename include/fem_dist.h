/*
 * fem_dist.h -- C-ABI of the multi-GPU stepping loop (libfem_dist.so): the halo exchange over RCCL and the
 * schedule that overlaps it with the interior tiles.
 *
 * No counterpart in the reference, which is single-GPU (src/main.cpp:194-205 picks one device; its timed loop is
 * main.cpp:558-562).  This is that loop for a mesh sharded along the partitioning_common partition ranges
 * (fem_shard_create, fem_hip.h section 5): k ghost rings let a rank take k steps between two exchanges; per exchange
 * the boundary state is packed on the device and sent with grouped ncclSend / ncclRecv (RCCL over xGMI) straight into
 * the ghost segment of the neighbours' state buffers, on a communication stream of its own, while the tiles that
 * need no ghost row run on the compute stream.  Plain C; no HIP or RCCL header is needed by the caller.
 * Kept in a library of its own so that libfem_hip.so does not depend on librccl.so.
 */
#ifndef FEM_DIST_H
#define FEM_DIST_H

#include "fem_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* text of the last failing call of this library on this thread (HIP / RCCL error strings) */
const char *fem_dist_error_detail(void);

/* devices of this process (hipGetDeviceCount / hipSetDevice for the calling thread), for callers without a HIP header */
int fem_dist_device_count(void);
int fem_dist_set_device(int device);

/* a non-blocking stream (hipStream_t as void*) for one rank's compute work */
int fem_dist_stream_create(void **stream);
void fem_dist_stream_destroy(void *stream);
int fem_dist_stream_synchronize(void *stream);

/* RCCL communicator of one rank.  id_out/id: the 128-byte ncclUniqueId made by rank 0 and handed to every rank by
 * the caller (a file, a socket, shared memory of one process ...).  comm is an ncclComm_t as void*. */
int fem_dist_rccl_unique_id(char id_out[128]);
int fem_dist_rccl_comm_create(const char id[128], int rank, int nranks, void **comm);
void fem_dist_rccl_comm_destroy(void *comm);

/* A world whose ranks are threads of ONE process: the exchange is a device-to-device copy between the ranks'
 * buffers (same GPU or peers) instead of RCCL.  Lets every rank of a sharded run share one GPU (tests, rehearsals). */
typedef struct fem_dist_world fem_dist_world;
int fem_dist_world_create(int nranks, fem_dist_world **out);
void fem_dist_world_destroy(fem_dist_world *w);

/* One rank's stepping state over its shard's plan (fem_plan_create_tiled / _shard with that shard; the plan and the
 * shard must outlive it).  Exactly one of comm (RCCL) / world (in-process) is given.  The plan's ghost rows must be
 * valid when the first step runs (fem_plan_set_state with the shard's loc2glob order does that).
 * Environment: FEM_OVERLAP = 0 (one launch per group, blocking exchange) | split | sync (tests) | default adaptive. */
typedef struct fem_dist fem_dist;
int fem_dist_create(fem_plan *plan, const fem_shard *shard, void *comm, fem_dist_world *world, fem_dist **out);
void fem_dist_destroy(fem_dist *d);
/* nsteps time steps, exchanging when the ghost rings are used up; asynchronous on `stream` (compute) and on the
 * rank's own communication stream.  The timed loop of main.cpp:558-562 for a shard. */
int fem_dist_run(fem_dist *d, int nsteps, double dt, const fem_model *model, void *stream);
/* Semi-implicit (IMEX) steps on the shard: fem_plan_run_imex_sharded (fem_hip.h) with the ghost exchange of the
 * right-hand side and of every search direction as grouped ncclSend / ncclRecv, and the two dot products of a CG
 * iteration as ncclAllReduce of two doubles, all stream-ordered on `stream` (no host synchronisation with RCCL).
 * The plan must be a sliced-ELL plan of the shard created with FEM_PLAN_IMPLICIT; one ghost ring is enough. */
int fem_dist_run_imex(fem_dist *d, int nsteps, double dt, const fem_model *model, int cg_iterations, void *stream);
/* The same steps on a shard's FUSED-TILE plan with the Chebyshev solver (fem_plan_run_imex_fused_sharded): halo
 * exchanges only -- right-hand side, then iterate and direction once per launch of ksteps iterations -- and no
 * all-reduce anywhere; wsum_max = half the whole mesh's fem_mesh_spectrum_bound (the same number on every rank). */
int fem_dist_run_imex_fused(fem_dist *d, int nsteps, double dt, const fem_model *model, double rel_tol, double wsum_max, void *stream);
/* makes `stream` wait for an exchange still in flight (call before reading the state back) */
int fem_dist_wait(fem_dist *d, void *stream);
/* counters since creation: exchanges started, groups run as one launch / interior-first */
void fem_dist_stats(const fem_dist *d, int *exchanges, int *merged_groups, int *split_groups);

#ifdef __cplusplus
}
#endif
#endif /* FEM_DIST_H */
