// device.hip -- gfx950 kernels and the device half of the C-ABI.
//
//   ref_step_kernel  : the reference's fused time step on the reference's own
//                      padded SoA layout (what `step` launches; reference
//                      src/FEM_common.cu:82-180), one workgroup per partition.
//   sell_step_kernel : the native time step on the sliced-ELL layout with
//                      folded gradient coefficients (pack.cpp), one wave64 per
//                      64-row slice, no atomics, no LDS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

// ---------------------------------------------------------------------------
// Reference-layout step (FEM_common.cu:82-166).  Differences from the reference
// kernel, none of them in the arithmetic: all loops are block-strided, so no
// thread leaves before a barrier and a partition may hold more faces or
// vertices than the block has threads (the reference returns early at :99-100
// and :138-139); the double accumulator uses the LDS f64 add instead of a CAS
// loop (:15-31).
// LDS: wg[size] double, then nvGrads[3*size], cvGrads[3*size] float = 32 B/vertex.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ref_step_kernel(
    const double2 *__restrict__ fn_src, double2 *__restrict__ fn_dst, const uint *__restrict__ fv,
    const double *__restrict__ fv_weights, uint fv_pitch, uint fvw_pitch, const uint *__restrict__ nbr,
    const double *__restrict__ vtxW, uint vw_pitch, const double *__restrict__ vertex_weights,
    uint vv_pitch, uint vws_pitch, uint vv_size, const float4 *__restrict__ grads, uint he_pitch,
    const uint *__restrict__ vertex_parts, const uint *__restrict__ block_face_count, double dt,
    ModelD m) {
  extern __shared__ double s_mem[];
  const uint p = blockIdx.x;
  const uint base = vertex_parts[p];
  const uint size = vertex_parts[p + 1] - base;
  double *s_wg = s_mem;
  float *s_nv = reinterpret_cast<float *>(s_wg + size);
  float *s_cv = s_nv + 3 * size;

  for (uint i = threadIdx.x; i < size; i += blockDim.x) {
    s_wg[i] = 0.0;
    s_nv[3 * i] = s_nv[3 * i + 1] = s_nv[3 * i + 2] = 0.0f;
    s_cv[3 * i] = s_cv[3 * i + 1] = s_cv[3 * i + 2] = 0.0f;
  }
  __syncthreads();

  /* face gradients, scattered to the owned corners */
  const uint nfaces = block_face_count[p];
  const size_t fvb = (size_t)p * 3 * fv_pitch, fwb = (size_t)p * 3 * fvw_pitch, heb = (size_t)p * 2 * he_pitch;
  for (uint t = threadIdx.x; t < nfaces; t += blockDim.x) {
    const uint i0 = fv[fvb + t], i1 = fv[fvb + fv_pitch + t], i2 = fv[fvb + 2 * (size_t)fv_pitch + t];
    const double2 v1 = fn_src[i2], a = fn_src[i0], b = fn_src[i1];
    const float d12x = (float)(a.x - v1.x), d12y = (float)(a.y - v1.y);
    const float d13x = (float)(b.x - v1.x), d13y = (float)(b.y - v1.y);
    const float4 g12 = grads[heb + t], g13 = grads[heb + he_pitch + t];
    const float nvx = g12.x * d12x + g13.x * d13x, nvy = g12.y * d12x + g13.y * d13x, nvz = g12.z * d12x + g13.z * d13x;
    const float cvx = g12.x * d12y + g13.x * d13y, cvy = g12.y * d12y + g13.y * d13y, cvz = g12.z * d12y + g13.z * d13y;
    const uint idx[3] = {i0, i1, i2};
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const uint li = idx[j] - base;              // unsigned wrap makes "below base" large
      if (li < size) {
        const double weight = fv_weights[fwb + (size_t)j * fvw_pitch + t];
        atomicAdd(&s_nv[3 * li], (float)(nvx * weight));
        atomicAdd(&s_nv[3 * li + 1], (float)(nvy * weight));
        atomicAdd(&s_nv[3 * li + 2], (float)(nvz * weight));
        atomicAdd(&s_cv[3 * li], (float)(cvx * weight));
        atomicAdd(&s_cv[3 * li + 1], (float)(cvy * weight));
        atomicAdd(&s_cv[3 * li + 2], (float)(cvz * weight));
        unsafeAtomicAdd(&s_wg[li], weight);
      }
    }
  }
  __syncthreads();

  /* laplacian, vertex gradients, update */
  const size_t nb = (size_t)p * (vv_size + 1) * vv_pitch, wb = (size_t)p * vv_size * vws_pitch;
  for (uint t = threadIdx.x; t < size; t += blockDim.x) {
    const uint i = base + t;
    const double2 s = fn_src[i];
    const double vW = vtxW[(size_t)p * vw_pitch + t];
    double lx = s.x * vW, ly = s.y * vW;
    const uint end = nbr[nb + t];
    for (uint j = 0; j < end; j++) {
      const uint nIdx = nbr[nb + (size_t)vv_pitch * (j + 1) + t];
      const double hW = vertex_weights[wb + (size_t)vws_pitch * j + t];
      const double2 f = fn_src[nIdx];
      lx += f.x * hW; ly += f.y * hW;
    }
    double dotP = 0;
    const double wg = s_wg[t];
    if (wg > 0) {
      const float w = (float)wg;
      dotP = (s_nv[3 * t] / w) * (s_cv[3 * t] / w) + (s_nv[3 * t + 1] / w) * (s_cv[3 * t + 1] / w) +
             (s_nv[3 * t + 2] / w) * (s_cv[3 * t + 2] / w);
    }
    const double dauN = m.D * lx - m.alpha * s.x * ly - m.alpha * dotP + m.S * m.r * s.x * (m.nMax - s.x);
    const double dauC = ly + m.S * (s.x / (1 + s.x) - s.y);
    double2 o;
    o.x = dt * dauN + s.x > 0 ? dt * dauN + s.x : 0.0;
    o.y = dt * dauC + s.y > 0 ? dt * dauC + s.y : 0.0;
    fn_dst[i] = o;
  }
}

// ---------------------------------------------------------------------------
// Native step.  One wave64 per slice; entry (k, lane) of a slice lives at
// slice_off + k*64 + lane, so every operator load is one fully coalesced
// 256-B (idx) or 1-KiB (coef, fp32) wave access; neighbour state is gathered
// from global memory (L1/L2-resident: a slice's neighbours are a few KB).
// Workgroups are renumbered so that consecutive slices run on one XCD and share
// its L2 (workgroups are dealt round-robin over the 8 XCDs).
// Algorithmic bytes per vertex (fp32, valence 6): 6*(4+16) operator + 8 state
// read + 8 state write = 136 B (+ gathers served on chip).
// ---------------------------------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<float> { typedef float2 t2; typedef float4 t4; };
template <> struct Vec<double> { typedef double2 t2; typedef double4 t4; };

// bn = base value the stored n is a deviation from; cden = 1/(1+bn); nMaxOff = nMax - bn
template <typename T> struct ModelT { T nMaxOff, D, r, alpha, S, dt, bn, cden; };

// Reaction + explicit Euler + clamp (FEM_common.cu:153-165) on deviation-form state.
// The buffers hold deviations (n - bn, c - bn/(1+bn)) from the homogeneous state: differences
// are unchanged, n(nMax - n) = n((nMax - bn) - dev) and n/(1+n) - c = dev_n/((1+n)(1+bn)) - dev_c
// are exact rewrites, and the update no longer rounds at the magnitude of n itself.
template <typename T>
__device__ __forceinline__ typename Vec<T>::t2 reaction_update(typename Vec<T>::t2 s, T lapn, T lapc, T gnx, T gny,
                                                               T gnz, T gcx, T gcy, T gcz, const ModelT<T> &m) {
  const T n = m.bn + s.x;
  const T dotP = gnx * gcx + gny * gcy + gnz * gcz;
  const T dauN = m.D * lapn - m.alpha * n * lapc - m.alpha * dotP + m.S * m.r * n * (m.nMaxOff - s.x);
  const T dauC = lapc + m.S * (s.x * m.cden / (1 + n) - s.y);
  const T un = m.dt * dauN + s.x, uc = m.dt * dauC + s.y;
  typename Vec<T>::t2 o;
  o.x = un + m.bn > 0 ? un : -m.bn;                 // max(0, n) of FEM_common.cu:164-165
  o.y = uc + m.bn * m.cden > 0 ? uc : -(m.bn * m.cden);
  return o;
}

// streaming (read-once) 16/32-byte load: keeps the operator rows from evicting the state lines in L2
__device__ __forceinline__ float4 ntload(const float4 *p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double4 ntload(const double4 *p) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d a = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p) + 1);
  return make_double4(a.x, a.y, b.x, b.y);
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void sell_step_kernel(
    const typename Vec<T>::t2 *__restrict__ src, typename Vec<T>::t2 *__restrict__ dst,
    const uint *__restrict__ slice_off, const uint *__restrict__ idx,
    const typename Vec<T>::t4 *__restrict__ coef, uint V, uint nslices, uint nblocks, ModelT<T> m) {
  typedef typename Vec<T>::t2 T2;
  typedef typename Vec<T>::t4 T4;
  // bijective XCD remap: blocks b, b+8, b+16, ... (one XCD) get consecutive slices
  const uint b = blockIdx.x, q = nblocks >> 3, r = nblocks & 7u, x = b & 7u;
  const uint lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  const uint slice = lb * 4 + (threadIdx.x >> 6);
  if (slice >= nslices) return;
  const uint lane = threadIdx.x & 63u;
  const uint off = __builtin_amdgcn_readfirstlane(slice_off[slice]);
  const uint end = __builtin_amdgcn_readfirstlane(slice_off[slice + 1]);
  const uint v = slice * 64 + lane;
  const uint vs = v < V ? v : V - 1;
  const T2 s = src[vs];
  T lapn = 0, lapc = 0, gnx = 0, gny = 0, gnz = 0, gcx = 0, gcy = 0, gcz = 0;
#pragma unroll 2
  for (uint e = off + lane; e < end; e += 64) {
    const uint j = NT ? __builtin_nontemporal_load(&idx[e]) : idx[e];     // read-once streams: do not evict the state
    const T4 c = NT ? ntload(&coef[e]) : coef[e];
    const T2 f = src[j];
    const T dn = f.x - s.x, dc = f.y - s.y;
    lapn += c.x * dn; lapc += c.x * dc;
    gnx += c.y * dn; gny += c.z * dn; gnz += c.w * dn;
    gcx += c.y * dc; gcy += c.z * dc; gcz += c.w * dc;
  }
  const T2 o = reaction_update<T>(s, lapn, lapc, gnx, gny, gnz, gcx, gcy, gcz, m);
  if (v < V) dst[v] = o;
}

// ---------------------------------------------------------------------------
// Fused k-step kernel.  One workgroup per tile (pack.cpp:packTileLayout): the tile's owned rows
// plus k rings of neighbours are staged in LDS (two ping-pong buffers of m[k] pairs), the
// operator rows of the first m[k-1] local rows are loaded ONCE into registers (thread t holds rows
// t, t+TPB, ...: R rows x W entries x (float4 coef + 16-bit local column id)), and the workgroup
// takes up to k time steps before writing its owned rows back: step s recomputes rows
// [0, m[k-1-s]), whose neighbours [0, m[k-s]) are valid from the step before.  HBM traffic per
// vertex-step drops from 136 B to about (1.45*108 + 1.6*8 + 8)/k B (44 B at k = 4, 1024-row tiles).
// ---------------------------------------------------------------------------
template <typename T, int TPB, int R, int W, int MINW, bool X, int PRO>
__global__ __launch_bounds__(TPB, MINW) void tile_step_kernel(
    const typename Vec<T>::t2 *__restrict__ src, typename Vec<T>::t2 *__restrict__ dst,
    const uint *__restrict__ meta, const uint *__restrict__ ghost, const uint *__restrict__ idx2,
    const typename Vec<T>::t4 *__restrict__ coef, const uint *__restrict__ xinfo, uint tile0, uint ntiles,
    uint maxLocal, uint maxExtra, int nsteps, ModelT<T> m) {
  typedef typename Vec<T>::t2 T2;
  typedef typename Vec<T>::t4 T4;
  extern __shared__ __align__(16) unsigned char tile_smem[];
  const uint b = blockIdx.x, q8 = ntiles >> 3, r8 = ntiles & 7u, x = b & 7u;
  const uint tile = tile0 + (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (b >> 3);   // tiles [tile0, tile0 + ntiles)
  const uint *mt = meta + 16 * (size_t)tile;
  const uint base = mt[0], goff = mt[1], roff = mt[2], rpad = mt[3], k = mt[4];
  const uint n0 = mt[5], mk = mt[5 + k];
  T2 *cur = reinterpret_cast<T2 *>(tile_smem);
  T2 *nxt = cur + maxLocal;
  // rows wider than W (irregular meshes): entries W, W+1, ... of all rows of the tile, in LDS
  T4 *xs_coef = reinterpret_cast<T4 *>(nxt + maxLocal);          // maxLocal is even: 16-B aligned
  uint *xs_off = reinterpret_cast<uint *>(xs_coef + maxExtra);
  uint *xs_info = xs_off + maxExtra;                             // [T*R] rows of the register tile
  uint *xs_ix = xs_info + TPB * R;                               // X: the packed column offsets live here, not in VGPRs
  constexpr bool IXL = true;                             // packed column offsets in LDS (frees W/2*R VGPRs)
  const uint xoff = mt[14], xcnt = X ? mt[15] : 0u;      // X = false: plan without wide rows
  const uint t = threadIdx.x;

  // tile-uniform (scalar) bases + 32-bit lane offsets keep the address arithmetic out of VGPRs
  const T4 *__restrict__ tcoef = coef + roff;
  const uint *__restrict__ tidx = idx2 + (roff >> 1);
  // Every load below is unconditional with a clamped index (no branch => no wait at a branch
  // merge), so the whole prologue is in flight at once: R*(W + W/2) operator loads, then the
  // ghost ids, then the state gathers; the first wait is the one in front of the LDS writes.
  T4 cf[R][W];
  uint ix[R][W / 2];
  constexpr int SU = 4;                                  // state slots per thread per pass
  uint gi[SU];
  auto loadGhostIds = [&](uint i0) {
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const uint i = i0 + t + TPB * u;
      const uint g = (i >= n0 && i < mk) ? i - n0 : 0;   // ghost[] carries one pad element
      gi[u] = PRO == 3 ? __builtin_nontemporal_load(&ghost[goff + g]) : ghost[goff + g];
    }
  };
  // Issue order = completion order (one in-order counter per wave): the ghost ids go first, so
  // the dependent state gathers can leave while the operator rows are still streaming in.
  // PRO (FEM_TILE_PROLOGUE, same-box A/B on the 16M sphere): 0 ids after the operator 219.0;
  // 1 ids first 220.2; 2 + operator rows as nontemporal loads 222.9; 3 + index streams
  // nontemporal 224.6-228.7 G DOF-it/s (the read-once streams stop evicting the state, which
  // is the only data with reuse between tiles and launches).
  if (PRO >= 1) loadGhostIds(0);
#pragma unroll
  for (int q = 0; q < R; q++) {
    const uint r = min(t + TPB * q, rpad - 1);         // rows in [rows, rpad) are zero padding
#pragma unroll
    for (int w = 0; w < W / 2; w++)                      // ids < 4096: both halves scale
      ix[q][w] = (PRO == 3 ? __builtin_nontemporal_load(&tidx[w * rpad + r]) : tidx[w * rpad + r]) * (uint)sizeof(T2);
  }
#pragma unroll
  for (int q = 0; q < R; q++) {
    const uint r = min(t + TPB * q, rpad - 1);
#pragma unroll
    for (int w = 0; w < W; w++) cf[q][w] = PRO >= 2 ? ntload(&tcoef[w * rpad + r]) : tcoef[w * rpad + r];
  }
  if (X && xcnt) {                                       // tile-uniform
    for (uint i = t; i < rpad; i += TPB) xs_info[i] = xinfo[xoff + i];      // first extra | count << 16 per row
    for (uint i = t; i < xcnt; i += TPB) {
      xs_coef[i] = tcoef[W * rpad + i];
      xs_off[i] = xinfo[xoff + rpad + i] * (uint)sizeof(T2);
    }
  }
  for (uint i0 = 0; i0 < mk; i0 += SU * TPB) {
    T2 sv[SU];
    if (PRO < 1 || i0 > 0) loadGhostIds(i0);
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const uint i = i0 + t + TPB * u;
      sv[u] = src[i < n0 ? base + i : gi[u]];
    }
    if (IXL && i0 == 0) {
#pragma unroll
      for (int q = 0; q < R; q++)
#pragma unroll
        for (int w = 0; w < W / 2; w++) xs_ix[w * (TPB * R) + t + TPB * q] = ix[q][w];
    }
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const uint i = i0 + t + TPB * u;
      if (i < mk) cur[i] = sv[u];
    }
  }
  __syncthreads();

  for (int s = 0; s < nsteps; s++) {
    const uint lim = mt[5 + k - 1 - s];
#pragma unroll
    for (int q = 0; q < R; q++) {
      const uint r = t + TPB * q;
      if (r < lim) {
        const T2 self = cur[r];
        T lapn = 0, lapc = 0, gnx = 0, gny = 0, gnz = 0, gcx = 0, gcy = 0, gcz = 0;
        if constexpr (IXL) {
#pragma unroll
          for (int w = 0; w < W / 2; w++) ix[q][w] = xs_ix[w * (TPB * R) + r];
        }
#pragma unroll
        for (int w = 0; w < W / 2; w++) asm volatile("" : "+v"(ix[q][w]));   // keep the unpacking in the loop, not in VGPRs
#pragma unroll
        for (int w = 0; w < W; w++) {
          const uint joff = (w & 1) ? (ix[q][w >> 1] >> 16) : (ix[q][w >> 1] & 0xffffu);   // byte offset in the buffer
          const T2 f = *reinterpret_cast<const T2 *>(reinterpret_cast<const unsigned char *>(cur) + joff);
          const T4 c = cf[q][w];
          const T dn = f.x - self.x, dc = f.y - self.y;
          lapn += c.x * dn; lapc += c.x * dc;
          gnx += c.y * dn; gny += c.z * dn; gnz += c.w * dn;
          gcx += c.y * dc; gcy += c.z * dc; gcz += c.w * dc;
        }
        if (X && xcnt) {                                 // entries W, W+1, ... of a wide row
          const uint xi = xs_info[r];
          for (uint e = xi & 0xffffu, ee = e + (xi >> 16); e < ee; e++) {
            const T2 f = *reinterpret_cast<const T2 *>(reinterpret_cast<const unsigned char *>(cur) + xs_off[e]);
            const T4 c = xs_coef[e];
            const T dn = f.x - self.x, dc = f.y - self.y;
            lapn += c.x * dn; lapc += c.x * dc;
            gnx += c.y * dn; gny += c.z * dn; gnz += c.w * dn;
            gcx += c.y * dc; gcy += c.z * dc; gcz += c.w * dc;
          }
        }
        nxt[r] = reaction_update<T>(self, lapn, lapc, gnx, gny, gnz, gcx, gcy, gcz, m);
      }
      __builtin_amdgcn_sched_barrier(0);      // one row at a time: keeps the live set within 128 VGPRs
    }
    __syncthreads();
    T2 *tmp = cur; cur = nxt; nxt = tmp;
  }
  for (uint i = t; i < n0; i += TPB) dst[base + i] = cur[i];
}


// ---------------------------------------------------------------------------
// Device assembly (initMeshStats on the GPU; reference src/mesh_function.cpp:70-397).
//   assemble_rows_*_kernel : one thread per operator row, straight from the vertex positions:
//     each entry names its neighbour and the two vertices opposite the edge (pack.cpp), so the
//     thread rebuilds the (at most two) faces on the edge, and the faces left of the row's
//     outgoing edges are all faces around the vertex once: lumped mass (sum of area/3), angle
//     sum, cotangent sums (A3, :219-271), angle-weighted hat gradients and the longest edge (A4)
//     are per-row sequential sums with the order fixed by the layout -- no atomics, no per-face
//     intermediate in HBM, bit-reproducible.  Writes (w, Cx, Cy, Cz) straight into the plan's
//     operator array.  Every face is recomputed by the rows that touch it (ALU is free here: the
//     kernel moves 16 B table + 16 B result per entry and gathers positions from L2).
//   assemble_faces_kernel : one thread per element, the reference's per-face/half-edge arrays
//     (A1 angles + cotangents :70-160, A2 area :162-216, A4 longest edge :310-350, A5 gradVec12/13
//     :351-397) as one 64-B record per face
//     {ang_a, ang_b, ang_c, area | cot_a, cot_b, cot_c, maxEdge | g12.xyz, 0 | g13.xyz, 0};
//     only run for fem_plan_face_records (checks, output), not needed by the operator.
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T dev_acos(T x);
template <> __device__ __forceinline__ float dev_acos<float>(float x) { return acosf(x); }
template <> __device__ __forceinline__ double dev_acos<double>(double x) { return acos(x); }
template <typename T> __device__ __forceinline__ T dev_tan(T x);
template <> __device__ __forceinline__ float dev_tan<float>(float x) { return tanf(x); }
template <> __device__ __forceinline__ double dev_tan<double>(double x) { return tan(x); }
template <typename T> __device__ __forceinline__ T dev_sqrt(T x);
template <> __device__ __forceinline__ float dev_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double dev_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T dev_rsqrt(T x);
template <> __device__ __forceinline__ float dev_rsqrt<float>(float x) { return __frsqrt_rn(x); }   // v_rsq_f32, 1 ulp
template <> __device__ __forceinline__ double dev_rsqrt<double>(double x) { return 1.0 / sqrt(x); }

template <typename T>
__device__ __forceinline__ void corner_angle(const T *p1, const T *p2, const T *p3, T &ang, T &cot) {
  // angle at p2 between p1 and p3; precedence of mesh_function.cpp:126: dot(v21, v23 / (|v21| |v23|))
  T v21[3], v23[3], n21 = 0, n23 = 0, d = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) { v21[k] = p1[k] - p2[k]; v23[k] = p3[k] - p2[k]; n21 += v21[k] * v21[k]; n23 += v23[k] * v23[k]; }
  const T sden = dev_sqrt(n21) * dev_sqrt(n23);
#pragma unroll
  for (int k = 0; k < 3; k++) d += v21[k] * (v23[k] / sden);
  ang = dev_acos(d);
  cot = (T)1 / dev_tan(ang);
}

template <typename T>
__global__ __launch_bounds__(256) void assemble_faces_kernel(const typename Vec<T>::t4 *__restrict__ pos,
                                                             const uint *__restrict__ tri0, const uint *__restrict__ tri1,
                                                             const uint *__restrict__ tri2, typename Vec<T>::t4 *__restrict__ rec,
                                                             uint F, double *__restrict__ blockMax) {
  typedef typename Vec<T>::t4 T4;
  const uint f = blockIdx.x * blockDim.x + threadIdx.x;
  T me = 0;
  if (f < F) {
    const T4 A = pos[tri0[f]], B = pos[tri1[f]], C = pos[tri2[f]];
    const T a[3] = {A.x, A.y, A.z}, b[3] = {B.x, B.y, B.z}, c[3] = {C.x, C.y, C.z};
    T4 r0, r1, r2, r3;
    corner_angle<T>(c, a, b, r0.x, r1.x);               // half-edge c->a ends at a
    corner_angle<T>(a, b, c, r0.y, r1.y);
    corner_angle<T>(b, c, a, r0.z, r1.z);
    // area = |(p1-p2) x (p3-p2)| / 2 with (p1,p2,p3) = (c,a,b)
    T u[3], v[3], e1[3], e2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { u[k] = c[k] - a[k]; v[k] = b[k] - a[k]; e1[k] = c[k] - b[k]; e2[k] = a[k] - b[k]; }
    const T cx = u[1] * v[2] - u[2] * v[1], cy = u[2] * v[0] - u[0] * v[2], cz = u[0] * v[1] - u[1] * v[0];
    const T area = dev_sqrt(cx * cx + cy * cy + cz * cz) / 2;
    r0.w = area;
    T nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
    const T nl = dev_sqrt(nx * nx + ny * ny + nz * nz);
    if (nl != 0) { nx /= nl; ny /= nl; nz /= nl; } else { nx = ny = nz = 0; }
    const T twoA = 2 * area;
    // gradVec12 = N x (pt1 - pt3) / 2A, (pt1, pt3) = (c, b);  gradVec13 = N x (pt2 - pt1) / 2A = N x (a - c) / 2A
    r2.x = (ny * e1[2] - nz * e1[1]) / twoA; r2.y = (nz * e1[0] - nx * e1[2]) / twoA; r2.z = (nx * e1[1] - ny * e1[0]) / twoA; r2.w = 0;
    r3.x = (ny * (-u[2]) - nz * (-u[1])) / twoA; r3.y = (nz * (-u[0]) - nx * (-u[2])) / twoA; r3.z = (nx * (-u[1]) - ny * (-u[0])) / twoA; r3.w = 0;
    const T lab = dev_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const T lbc = dev_sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    const T lca = dev_sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    me = fmax(lab, fmax(lbc, lca));
    r1.w = me;
    rec[4 * (size_t)f] = r0; rec[4 * (size_t)f + 1] = r1; rec[4 * (size_t)f + 2] = r2; rec[4 * (size_t)f + 3] = r3;
  }
  // A4: longest edge -> wave shuffle reduction, one partial per workgroup (a single-address atomic
  // per wave would serialise ~F/64 atomics at the memory side); max_reduce_kernel finishes it
  __shared__ double s_part[4];
  double m = (double)me;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) blockMax[blockIdx.x] = fmax(fmax(s_part[0], s_part[1]), fmax(s_part[2], s_part[3]));
}

__global__ __launch_bounds__(1024) void max_reduce_kernel(const double *__restrict__ part, uint n, double *__restrict__ out) {
  __shared__ double s_part[16];
  double m = 0;
  for (uint i = threadIdx.x; i < n; i += 1024) m = fmax(m, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; i++) m = fmax(m, s_part[i]);
    *out = m;
  }
}

constexpr int kTileWDev = 6;       // entries per row of the tile layout (= kTileW)

// One corner of the fan around v: face (v, p, q), counter-clockwise, e1 = p - v, e2 = q - v.
//   c = e1 x e2 (|c| = 2 area), angle at v = acos(e1.e2 / (|e1||e2|)),
//   cot at q = (|e2|^2 - e1.e2) / |c|, cot at p = (|e1|^2 - e1.e2) / |c|,
//   grad phi_p = e2 x c / |c|^2, grad phi_q = c x e1 / |c|^2
// (hat gradient = normal x opposite edge / 2 area).  Same quantities as mesh_function.cpp:70-397
// (A1 angles/cotangents, A2 areas, A5 gradVec12/13), per row instead of per face.
template <typename T> struct FanFace { T cx, cy, cz, c2, d, l1, l2; };
template <typename T>
__device__ __forceinline__ FanFace<T> fan_face(const T *e1, const T *e2) {
  FanFace<T> f;
  f.cx = e1[1] * e2[2] - e1[2] * e2[1]; f.cy = e1[2] * e2[0] - e1[0] * e2[2]; f.cz = e1[0] * e2[1] - e1[1] * e2[0];
  f.c2 = f.cx * f.cx + f.cy * f.cy + f.cz * f.cz;
  f.d = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
  f.l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
  f.l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
  return f;
}

// One operator row straight from the positions: entries e(k) = first + k * stride for k < SPLIT,
// xfirst + (k - SPLIT) beyond (the tile layout's extras of a wide row), k < width <= W.
// Per entry (v -> j): face (v, j, o1) on the left and (v, o2, j) on the right give the cotangents
// opposite the edge (A3, mesh_function.cpp:219-271) and the angle-weighted hat gradients of j
// (mesh_function.h:35-110); the faces left of v's outgoing half-edges are all faces around v once,
// so the lumped mass and the angle sum are row sums of the same quantities.  Returns the longest
// edge at v (A4, :310-350).
template <typename T>
__device__ __forceinline__ typename Vec<T>::t4 entry_terms(const typename Vec<T>::t4 *__restrict__ pos,
                                                           const typename Vec<T>::t4 &P, uint vg, const uint4 &t, T &area3,
                                                           T &theta, T &maxe2) {
  typedef typename Vec<T>::t4 T4;
  const bool real = t.w != 0u, has1 = real && t.y != ~0u, has2 = real && t.z != ~0u;
  const T4 J = pos[real ? t.x : vg], O1 = pos[has1 ? t.y : vg], O2 = pos[has2 ? t.z : vg];
  const T ej[3] = {J.x - P.x, J.y - P.y, J.z - P.z};
  const T e1[3] = {O1.x - P.x, O1.y - P.y, O1.z - P.z};
  const T e2[3] = {O2.x - P.x, O2.y - P.y, O2.z - P.z};
  T4 a;
  a.x = a.y = a.z = a.w = 0;
  if (has1) {                                          // face (v, j, o1): p = j, q = o1
    const FanFace<T> f = fan_face<T>(ej, e1);
    const T rcn = dev_rsqrt(f.c2);                     // 1 / |c|
    const T ang = dev_acos(f.d * dev_rsqrt(f.l1 * f.l2));
    a.x += (f.l2 - f.d) * rcn;                         // cot at o1
    const T s = ang * rcn * rcn;                       // angle * grad phi_j, grad phi_p = e2 x c / |c|^2
    a.y += s * (e1[1] * f.cz - e1[2] * f.cy); a.z += s * (e1[2] * f.cx - e1[0] * f.cz); a.w += s * (e1[0] * f.cy - e1[1] * f.cx);
    theta += ang; area3 += f.c2 * rcn / 6;             // area / 3 = |c| / 6
  }
  if (has2) {                                          // face (v, o2, j): p = o2, q = j
    const FanFace<T> f = fan_face<T>(e2, ej);
    const T rcn = dev_rsqrt(f.c2);
    const T ang = dev_acos(f.d * dev_rsqrt(f.l1 * f.l2));
    a.x += (f.l1 - f.d) * rcn;                         // cot at o2
    const T s = ang * rcn * rcn;                       // grad phi_q = c x e1 / |c|^2
    a.y += s * (f.cy * e2[2] - f.cz * e2[1]); a.z += s * (f.cz * e2[0] - f.cx * e2[2]); a.w += s * (f.cx * e2[1] - f.cy * e2[0]);
  }
  if (real) maxe2 = fmax(maxe2, ej[0] * ej[0] + ej[1] * ej[1] + ej[2] * ej[2]);
  return a;
}

template <typename T>
__device__ __forceinline__ typename Vec<T>::t4 entry_scale(const typename Vec<T>::t4 &a, T area3, T theta) {
  typename Vec<T>::t4 o;
  o.x = area3 > 0 ? a.x / (2 * area3) : (T)0;
  if (theta > 0) { o.y = a.y / theta; o.z = a.z / theta; o.w = a.w / theta; } else { o.y = o.z = o.w = 0; }
  return o;
}

// W <= 12: fully unrolled, unnormalised terms in registers.  W = 0: any width, a plain loop that
// parks the unnormalised terms in the output array and rescales them in a second sweep.
template <typename T, int W, int SPLIT = W>
__device__ __forceinline__ T assemble_row(const typename Vec<T>::t4 *__restrict__ pos, const uint4 *__restrict__ topo,
                                          typename Vec<T>::t4 *__restrict__ coef, uint vg, size_t first, size_t stride,
                                          uint width, size_t xfirst = 0) {
  typedef typename Vec<T>::t4 T4;
  auto at = [&](int k) { return k < SPLIT ? first + k * stride : xfirst + (size_t)(k - SPLIT); };
  const T4 P = pos[vg];
  T area3 = 0, theta = 0, maxe2 = 0;
  if constexpr (W > 0) {
    uint4 t[W];
#pragma unroll
    for (int k = 0; k < W; k++) t[k] = (uint)k < width ? topo[at(k)] : make_uint4(0u, ~0u, ~0u, 0u);
    T4 acc[W];
#pragma unroll
    for (int k = 0; k < W; k++) acc[k] = entry_terms<T>(pos, P, vg, t[k], area3, theta, maxe2);
#pragma unroll
    for (int k = 0; k < W; k++)
      if ((uint)k < width) coef[at(k)] = entry_scale<T>(acc[k], area3, theta);
  } else {
    for (int k = 0; k < (int)width; k++) coef[at(k)] = entry_terms<T>(pos, P, vg, topo[at(k)], area3, theta, maxe2);
    for (int k = 0; k < (int)width; k++) coef[at(k)] = entry_scale<T>(coef[at(k)], area3, theta);
  }
  return dev_sqrt(maxe2);
}

template <typename T>
__device__ __forceinline__ T assemble_row_wide(const typename Vec<T>::t4 *__restrict__ pos, const uint4 *__restrict__ topo,
                                               typename Vec<T>::t4 *__restrict__ coef, uint vg, size_t first, size_t stride,
                                               uint width, size_t xfirst) {
  return assemble_row<T, 0, kTileWDev>(pos, topo, coef, vg, first, stride, width, xfirst);
}

// longest edge: wave shuffle reduction, one partial per workgroup (max_reduce_kernel finishes it)
__device__ __forceinline__ void block_max_256(double m, double *__restrict__ blockMax) {
  __shared__ double s_part[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) blockMax[blockIdx.x] = fmax(fmax(s_part[0], s_part[1]), fmax(s_part[2], s_part[3]));
}

template <typename T>
__global__ __launch_bounds__(256) void assemble_rows_sell_kernel(const typename Vec<T>::t4 *__restrict__ pos,
                                                                 const uint4 *__restrict__ topo, const uint *__restrict__ slice_off,
                                                                 const uint *__restrict__ rowg,
                                                                 typename Vec<T>::t4 *__restrict__ coef, uint nslices, uint V,
                                                                 double *__restrict__ blockMax) {
  const uint slice = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  T me = 0;
  if (slice < nslices) {
    const uint off = slice_off[slice], width = (slice_off[slice + 1] - off) >> 6;
    const uint row = min(slice * 64 + lane, V - 1);    // rows >= V: padding entries only (flag 0)
    const uint vg = rowg[row];
    if (width <= 6) me = assemble_row<T, 6>(pos, topo, coef, vg, (size_t)off + lane, 64, width);
    else if (width <= 12) me = assemble_row<T, 12>(pos, topo, coef, vg, (size_t)off + lane, 64, width);
    else me = assemble_row<T, 0, (1 << 30)>(pos, topo, coef, vg, (size_t)off + lane, 64, width);   // no split: all entries strided
  }
  block_max_256((double)me, blockMax);
}

template <typename T>
__global__ __launch_bounds__(256) void assemble_rows_tile_kernel(const typename Vec<T>::t4 *__restrict__ pos,
                                                                 const uint4 *__restrict__ topo, const uint *__restrict__ meta,
                                                                 const uint *__restrict__ ghost, const uint *__restrict__ xinfo,
                                                                 const uint *__restrict__ rowg,
                                                                 typename Vec<T>::t4 *__restrict__ coef, uint W,
                                                                 double *__restrict__ blockMax) {
  const uint *mt = meta + 16 * (size_t)blockIdx.x;
  const uint base = mt[0], goff = mt[1], roff = mt[2], rpad = mt[3], n0 = mt[5], rows = mt[5 + mt[4] - 1];
  const uint xoff = mt[14], xcnt = mt[15];
  T me = 0;
  for (uint r = threadIdx.x; r < rows; r += blockDim.x) {
    const uint vg = rowg[r < n0 ? base + r : ghost[goff + (r - n0)]];
    const uint xi = xcnt ? xinfo[xoff + r] : 0u, nx = xi >> 16;
    const size_t first = (size_t)roff + r, xfirst = (size_t)roff + (size_t)W * rpad + (xi & 0xffffu);
    T e;
    if (nx == 0) e = assemble_row<T, kTileWDev>(pos, topo, coef, vg, first, rpad, W);
    else if (nx <= 6) e = assemble_row<T, 12, kTileWDev>(pos, topo, coef, vg, first, rpad, W + nx, xfirst);
    else e = assemble_row_wide<T>(pos, topo, coef, vg, first, rpad, W + nx, xfirst);
    me = fmax(me, e);
  }
  block_max_256((double)me, blockMax);
}

// seeded smooth radial displacement of the base positions (BASELINE config 5: deforming mesh)
template <typename T>
__global__ __launch_bounds__(256) void deform_kernel(const typename Vec<T>::t4 *__restrict__ base,
                                                     typename Vec<T>::t4 *__restrict__ pos, uint V, T amp, T q, T phase) {
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  typename Vec<T>::t4 p = base[i];
  const T s = (T)1 + amp * sin(q * p.x + phase) * cos(q * p.y - (T)0.5 * phase) * sin(q * p.z + (T)0.25 * phase);
  p.x *= s; p.y *= s; p.z *= s;
  pos[i] = p;
}


// ---------------------------------------------------------------------------
// Result output (formatKernel + the two thrust reductions of format(), reference
// src/FEM_common.cu:183-230, without GL): min/max of n over the vertices, then
// RGBA = (n^, 1 - n^, 1 - n^, 1) with n^ = (n - min) / (max - min).
// ---------------------------------------------------------------------------
template <typename T2>
__global__ __launch_bounds__(256) void minmax_kernel(const T2 *__restrict__ state, uint V, double *__restrict__ part) {
  __shared__ double s_lo[4], s_hi[4];
  double lo = 1e300, hi = -1e300;
  for (uint i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
    const double x = (double)state[i].x;
    lo = fmin(lo, x); hi = fmax(hi, x);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = fmin(fmin(s_lo[0], s_lo[1]), fmin(s_lo[2], s_lo[3]));
    part[2 * blockIdx.x + 1] = fmax(fmax(s_hi[0], s_hi[1]), fmax(s_hi[2], s_hi[3]));
  }
}

__global__ __launch_bounds__(256) void minmax_final_kernel(const double *__restrict__ part, uint n, double *__restrict__ out) {
  __shared__ double s_lo[4], s_hi[4];
  double lo = 1e300, hi = -1e300;
  for (uint i = threadIdx.x; i < n; i += 256) { lo = fmin(lo, part[2 * i]); hi = fmax(hi, part[2 * i + 1]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = fmin(fmin(s_lo[0], s_lo[1]), fmin(s_lo[2], s_lo[3]));
    out[1] = fmax(fmax(s_hi[0], s_hi[1]), fmax(s_hi[2], s_hi[3]));
  }
}

template <typename T2>
__global__ __launch_bounds__(256) void format_kernel(const T2 *__restrict__ state, uint V, const double *__restrict__ mm,
                                                     double4 *__restrict__ rgba) {
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  const double n = ((double)state[i].x - mm[0]) / (mm[1] - mm[0]);
  rgba[i] = make_double4(n, 1 - n, 1 - n, 1.0);
}

// halo pack: dst[i] = state[send_idx[i]] (8 or 16 B per lane, coalesced stores)
template <typename T2>
__global__ __launch_bounds__(256) void pack_halo_kernel(const T2 *__restrict__ state, const uint *__restrict__ send_idx,
                                                        T2 *__restrict__ dst, uint n) {
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = state[send_idx[i]];
}

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
// waves per SIMD (sets the VGPR budget: 4 -> 128, 3 -> 168).  FEM_TILE_CONFIG picks one (tuning knob;
// 0 = 512 x 2 is the measured best: no spills, 2 workgroups per CU, 1024 register-resident rows per tile).
constexpr int kTileW = 6;
constexpr fem_uint kTileMaxLocal = 4096;
constexpr fem_uint kTileMaxExtra = 1536;  // LDS-resident overflow entries per tile (rows with more than kTileW neighbours)
struct TileCfg { int tpb, r, minw; };
constexpr TileCfg kTileCfgs[] = {{512, 2, 4}, {256, 4, 3}, {512, 3, 4}, {256, 3, 4}, {256, 2, 4}, {1024, 2, 4}};
constexpr int kNumTileCfgs = sizeof(kTileCfgs) / sizeof(kTileCfgs[0]);


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
  int ksteps = 0, tcfg = 0;
  fem_uint ntiles = 0, maxLocal = 0, maxExtra = 0;
  fem_uint nInterior = 0, nBoundary = 0, nGhost = 0;   // tile classes of a shard's plan, in this order
  uint64_t tile_rows = 0, tile_entries = 0;
  uint64_t tile_bytes = 0;
  DevBuf<fem_uint> tmeta, tghost, tidx2, txinfo;
  void *tcoef = nullptr;
  // device assembly (FEM_PLAN_DEVICE_ASSEMBLY): connectivity SoA, positions, per-face records, entry tables
  bool can_assemble = false;
  fem_uint F = 0, Vg = 0;
  DevBuf<fem_uint> tri0, tri1, tri2, topo, rowg;
  void *pos = nullptr, *pos0 = nullptr, *rec = nullptr;      // T4[Vg], T4[Vg], T4[4F]
  double *maxedge = nullptr, *blockmax = nullptr;
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
  fem_uint nb;
  if (p->ksteps > 0) {
    nb = p->ntiles;
    hipLaunchKernelGGL(assemble_rows_tile_kernel<T>, dim3(nb), dim3(256), 0, st, (const T4 *)p->pos, (const uint4 *)p->topo.p,
                       p->tmeta.p, p->tghost.p, p->txinfo.p, p->rowg.p, (T4 *)p->tcoef, (fem_uint)kTileW, p->blockmax);
  } else {
    nb = p->nblocks;
    hipLaunchKernelGGL(assemble_rows_sell_kernel<T>, dim3(nb), dim3(256), 0, st, (const T4 *)p->pos, (const uint4 *)p->topo.p,
                       p->slice_off.p, p->rowg.p, (T4 *)p->coef, p->nslices, p->V, p->blockmax);
  }
  hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(1024), 0, st, (const double *)p->blockmax, nb, p->maxedge);
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
  // One variant for regular and irregular meshes: the build with the wide-row loop measured 4 %
  // FASTER on the regular 16M sphere than the one without it (same-box A/B, 213.5 vs 204.3 G
  // DOF-it/s), so the X = false instantiation is not dispatched.
  static const int pro = getenv("FEM_TILE_PROLOGUE") ? atoi(getenv("FEM_TILE_PROLOGUE")) : 3;   // tuning knob; 3 measured best
  auto kern = pro == 3 ? tile_step_kernel<T, TPB, R, kTileW, MINW, true, 3> : pro == 2 ? tile_step_kernel<T, TPB, R, kTileW, MINW, true, 2>
                       : (pro == 1 ? tile_step_kernel<T, TPB, R, kTileW, MINW, true, 1> : tile_step_kernel<T, TPB, R, kTileW, MINW, true, 0>);
  const size_t lds = 2 * (size_t)p->maxLocal * sizeof(T2) + (size_t)p->maxExtra * (sizeof(T4) + 4) +
                     (size_t)TPB * R * 4 * (1 + kTileW / 2);
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

extern "C" {

int fem_last_error(void) { return fem::g_last_error; }
const char *fem_error_detail(void) { return fem::g_error_detail.c_str(); }

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
  const size_t lds = (size_t)max_part_vertices * 32;         // main.cpp:369
  if (max_part_vertices == 0 || lds > 160 * 1024)
    return fail(FEM_ERR_LIMIT, "fem_step: partition too large for the LDS accumulators (<= 5120 vertices)");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  ModelD m = {model->nMax, model->D, model->r, model->alpha, model->S};
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)ref_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(ref_step_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream,
                     (const double2 *)fn_src, (double2 *)fn_dst, fv, fv_weights,
                     fv_pitchInBytes / (fem_uint)sizeof(fem_uint), fvw_pitchInBytes / (fem_uint)sizeof(double),
                     nbr, vtxW, vw_pitchInBytes / (fem_uint)sizeof(double), vertex_weights,
                     vv_pitchInBytes / (fem_uint)sizeof(fem_uint), vws_pitchInBytes / (fem_uint)sizeof(double),
                     vv_size, (const float4 *)grads, he_pitchInBytes / (fem_uint)sizeof(fem_float4), parts_n,
                     block_face_count, dt, m);
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
  p->tri0.release(); p->tri1.release(); p->tri2.release(); p->topo.release(); p->rowg.release();
  for (void *q : {p->pos, p->pos0, p->rec, (void *)p->maxedge, (void *)p->blockmax}) if (q) (void)hipFree(q);
  if (p->tcoef) (void)hipFree(p->tcoef);
  if (p->coef) (void)hipFree(p->coef);
  for (void *s : p->state) if (s) (void)hipFree(s);
  delete p;
}

static int planCreate(const fem_mesh *ordered, const fem::Shard *shard, int precision, const fem_model *model,
                      int ksteps, int n, const fem_uint *node_parts, unsigned flags, fem_plan **out) {
  if (!ordered || !out || (precision != FEM_PREC_F32 && precision != FEM_PREC_F64) || ksteps < 0 || ksteps > 8)
    return fail(FEM_ERR_INVALID, "fem_plan_create: bad argument");
  if (ksteps > 0 && (!node_parts || n < 1)) return fail(FEM_ERR_INVALID, "fem_plan_create_tiled: partition ranges missing");
  fem_model dm;
  if (!model) { fem_default_model(&dm); model = &dm; }
  if (!(model->nMax > -1.0)) return fail(FEM_ERR_INVALID, "fem_plan_create: nMax must be > -1");
  fem::RowOp op;
  int rc = shard ? fem::buildRowOp(ordered->mesh, ordered->stats, op, shard->loc2glob.data(), (fem::uint)shard->loc2glob.size())
                 : fem::buildRowOp(ordered->mesh, ordered->stats, op);
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
    tcfg = cfgEnv ? atoi(cfgEnv) : 0;
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
  p->base_n = model->nMax; p->base_c = model->nMax / (1 + model->nMax);
  p->precision = precision; p->V = op.V;
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
    ok(p->tmeta.alloc(TL.meta.size())) && ok(p->tmeta.upload(TL.meta.data())) &&
    ok(p->tghost.alloc(TL.ghost.size())) && (TL.ghost.empty() || ok(p->tghost.upload(TL.ghost.data()))) &&
    ok(p->tidx2.alloc(TL.idx2.size())) && ok(p->tidx2.upload(TL.idx2.data())) &&
    ok(p->txinfo.alloc(TL.xinfo.size())) && ok(p->txinfo.upload(TL.xinfo.data()));
    uploadCoef(&p->tcoef, TL.coef);
  } else {
    p->nslices = L.nslices; p->nblocks = (L.nslices + 3) / 4; p->entries = L.idx.size();
    ok(p->slice_off.alloc(L.slice_off.size())) && ok(p->slice_off.upload(L.slice_off.data())) &&
    ok(p->idx.alloc(L.idx.size())) && ok(p->idx.upload(L.idx.data()));
    uploadCoef(&p->coef, L.coef);
  }
  ok(hipMalloc(&p->state[0], 2 * sb * p->V)) && ok(hipMalloc(&p->state[1], 2 * sb * p->V));
  if (e == hipSuccess && (flags & FEM_PLAN_DEVICE_ASSEMBLY)) {
    const fem::SimpleTriMesh &m = ordered->mesh;
    p->F = m.n_faces(); p->Vg = m.n_vertices(); p->can_assemble = true;
    std::vector<fem_uint> t0(p->F), t1(p->F), t2(p->F);
    for (fem_uint f = 0; f < p->F; f++) { t0[f] = m.faces[3 * (size_t)f]; t1[f] = m.faces[3 * (size_t)f + 1]; t2[f] = m.faces[3 * (size_t)f + 2]; }
    const std::vector<fem_uint> &topo = ksteps > 0 ? TL.topo : L.topo;
    ok(p->tri0.alloc(p->F)) && ok(p->tri0.upload(t0.data())) && ok(p->tri1.alloc(p->F)) && ok(p->tri1.upload(t1.data())) &&
    ok(p->tri2.alloc(p->F)) && ok(p->tri2.upload(t2.data())) && ok(p->topo.alloc(topo.size())) && ok(p->topo.upload(topo.data())) &&
    ok(p->rowg.alloc(op.rowg.size())) && ok(p->rowg.upload(op.rowg.data())) &&
    ok(hipMalloc(&p->pos, 4 * sb * p->Vg)) && ok(hipMalloc(&p->pos0, 4 * sb * p->Vg)) &&
    ok(hipMalloc((void **)&p->maxedge, 8)) && ok(hipMemset(p->maxedge, 0, 8)) &&
    ok(hipMalloc((void **)&p->blockmax, 8 * (size_t)(std::max<size_t>(ksteps > 0 ? TL.ntiles : (L.nslices + 3) / 4, 1))));
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
  return planCreate(ordered, nullptr, precision, model, 0, 0, nullptr, 0, out);
}

int fem_plan_create_ex(const fem_mesh *ordered, int n, const fem_uint *node_parts, const fem_shard *shard,
                       int precision, const fem_model *model, int ksteps, unsigned flags, fem_plan **out) {
  int rc = planCreate(ordered, shard ? &shard->s : nullptr, precision, model, ksteps, n, node_parts, flags, out);
  if (rc == FEM_OK && (flags & FEM_PLAN_DEVICE_ASSEMBLY)) {
    rc = fem_plan_set_points(*out, ordered->mesh.points.data());
    if (rc != FEM_OK) { fem_plan_destroy(*out); *out = nullptr; }
  }
  return rc;
}

int fem_plan_create_tiled(const fem_mesh *ordered, int n, const fem_uint *node_parts, const fem_shard *shard,
                          int precision, const fem_model *model, int ksteps, fem_plan **out) {
  if (ksteps < 1) return fail(FEM_ERR_INVALID, "fem_plan_create_tiled: ksteps must be 1..8");
  return planCreate(ordered, shard ? &shard->s : nullptr, precision, model, ksteps, n, node_parts, 0, out);
}

int fem_plan_create_shard(const fem_mesh *ordered, const fem_shard *shard, int precision, const fem_model *model,
                          fem_plan **out) {
  if (!shard) return fail(FEM_ERR_INVALID, "fem_plan_create_shard: null shard");
  return planCreate(ordered, &shard->s, precision, model, 0, 0, nullptr, 0, out);
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
  if (!p || !nFn || !cFn) return fail(FEM_ERR_INVALID, "fem_plan_set_state: null");
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
}

int fem_plan_get_state(fem_plan *p, double *nFn, double *cFn) {
  if (!p || !nFn || !cFn) return fail(FEM_ERR_INVALID, "fem_plan_get_state: null");
  HIP_TRY(hipDeviceSynchronize());
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
  switch (p->tcfg) { FEM_TILE_CASE(0) FEM_TILE_CASE(1) FEM_TILE_CASE(2) FEM_TILE_CASE(3) FEM_TILE_CASE(4) FEM_TILE_CASE(5) }
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

int fem_plan_tile_classes(const fem_plan *p, fem_uint *n_interior, fem_uint *n_boundary, fem_uint *n_ghost) {
  if (!p || p->ksteps <= 0) return fail(FEM_ERR_INVALID, "fem_plan_tile_classes: not a tiled plan");
  if (n_interior) *n_interior = p->nInterior;
  if (n_boundary) *n_boundary = p->nBoundary;
  if (n_ghost) *n_ghost = p->nGhost;
  return FEM_OK;
}

int fem_plan_run(fem_plan *p, int nsteps, double dt, const fem_model *model, void *stream) {
  if (!p || nsteps < 0) return fail(FEM_ERR_INVALID, "fem_plan_run: bad argument");
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
    static const bool nt = !(getenv("FEM_SELL_NT") && atoi(getenv("FEM_SELL_NT")) == 0);
    if (p->precision == FEM_PREC_F32)
      hipLaunchKernelGGL((nt ? sell_step_kernel<float, true> : sell_step_kernel<float, false>), dim3(p->nblocks), dim3(256), 0, (hipStream_t)stream,
                         (const float2 *)src, (float2 *)dst, p->slice_off.p, p->idx.p, (const float4 *)p->coef,
                         p->V, p->nslices, p->nblocks, mf);
    else
      hipLaunchKernelGGL((nt ? sell_step_kernel<double, true> : sell_step_kernel<double, false>), dim3(p->nblocks), dim3(256), 0, (hipStream_t)stream,
                         (const double2 *)src, (double2 *)dst, p->slice_off.p, p->idx.p, (const double4 *)p->coef,
                         p->V, p->nslices, p->nblocks, md);
    p->cur ^= 1;
  }
  HIP_TRY(hipGetLastError());
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
  if (!p || !points || !p->can_assemble) return fail(FEM_ERR_INVALID, "fem_plan_set_points: plan was not created with FEM_PLAN_DEVICE_ASSEMBLY");
  return p->precision == FEM_PREC_F32 ? setPoints<float>(p, points) : setPoints<double>(p, points);
}

int fem_plan_assemble(fem_plan *p, void *stream) {
  if (!p || !p->can_assemble) return fail(FEM_ERR_INVALID, "fem_plan_assemble: plan was not created with FEM_PLAN_DEVICE_ASSEMBLY");
  if (p->precision == FEM_PREC_F32) launchAssemble<float>(p, (hipStream_t)stream);
  else launchAssemble<double>(p, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return FEM_OK;
}

int fem_plan_deform(fem_plan *p, double amplitude, double wavenumber, double phase, void *stream) {
  if (!p || !p->can_assemble) return fail(FEM_ERR_INVALID, "fem_plan_deform: plan was not created with FEM_PLAN_DEVICE_ASSEMBLY");
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
  if (!p || !p->can_assemble || !rec16F) return fail(FEM_ERR_INVALID, "fem_plan_face_records: bad argument");
  HIP_TRY(p->precision == FEM_PREC_F32 ? launchFaceRecords<float>(p) : launchFaceRecords<double>(p));
  return p->precision == FEM_PREC_F32 ? downloadAs<float>(p->rec, 16 * (size_t)p->F, rec16F)
                                      : downloadAs<double>(p->rec, 16 * (size_t)p->F, rec16F);
}

uint64_t fem_plan_coef_count(const fem_plan *p) {
  if (!p) return 0;
  return p->ksteps > 0 ? p->tile_entries * 4 : (uint64_t)p->entries * 4;
}

int fem_plan_download_coef(fem_plan *p, double *out) {
  if (!p || !out) return fail(FEM_ERR_INVALID, "fem_plan_download_coef: null");
  const void *dev = p->ksteps > 0 ? p->tcoef : p->coef;
  const size_t n = fem_plan_coef_count(p);
  return p->precision == FEM_PREC_F32 ? downloadAs<float>(dev, n, out) : downloadAs<double>(dev, n, out);
}


// ---- result output ------------------------------------------------------------------------------------

int fem_plan_format(fem_plan *p, fem_uint nverts, double *rgba, unsigned char *rgb, double *nmin, double *nmax) {
  if (!p || nverts == 0 || nverts > p->V) return fail(FEM_ERR_INVALID, "fem_plan_format: bad vertex count");
  const fem_uint V = nverts, nb = std::min<fem_uint>((V + 255) / 256, 1024);
  double *part = nullptr, *mm = nullptr;
  double4 *out = nullptr;
  hipError_t e = hipMalloc((void **)&part, 16 * (size_t)nb);
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
        rgb[3 * i + k] = (unsigned char)(c < 0 ? 0 : (c > 255 ? 255 : c));
      }
  if (nmin) *nmin = hmm[0] + p->base_n;     // the buffers hold deviations from base_n
  if (nmax) *nmax = hmm[1] + p->base_n;
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

uint64_t fem_plan_operator_bytes(const fem_plan *p) {
  if (!p) return 0;
  const size_t sb = p->precision == FEM_PREC_F32 ? 4 : 8;
  if (p->ksteps > 0) return p->tile_bytes;
  return (uint64_t)p->entries * (4 + 4 * sb) + (uint64_t)(p->nslices + 1) * 4;
}

}  // extern "C"
