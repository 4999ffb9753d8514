// kernels_step.hpp -- the time-step kernels: ref_step_kernel (reference layout, what `step` launches), sell_step_kernel (sliced ELL), tile_step_kernel (fused k steps)
// Included by device.hip inside its anonymous namespace (one translation unit: the kernels are
// launched from the C-ABI functions there).
#pragma once

// ---------------------------------------------------------------------------
// Reference-layout step (FEM_common.cu:82-166): the reference's arithmetic on the reference's own
// arrays, restructured for this chip.  The reference scatters every face's weighted gradients to
// its corners with 21 LDS atomics per face (6 float adds + 1 CAS-loop double add per corner,
// :118-132); measured here that scatter was 3/4 of the kernel's time.  Instead each workgroup
// builds the corner lists of its vertices in LDS (one integer atomic per owned corner; a vertex has
// at most vv_size incident faces, so the lists have fixed slots), parks the per-face gradients in
// LDS, and every vertex then SUMS its own corners in ascending face-slot order: 3 integer atomics
// per face instead of 21 floating-point ones, the same products ((float)(grad * weight), :122-127) added in float, the weight sum in
// double, and -- unlike the reference -- a bit-reproducible result.
// Further differences, none in the arithmetic: all loops are block-strided, so no thread leaves
// before a barrier and a partition may hold more faces or vertices than the block has threads
// (the reference returns early at :99-100 and :138-139); partitions are dealt to the XCDs in
// consecutive runs.
// LDS: cur[size], ent[size*vv_size] uint; fg[6*fpitch] float  (4*size*(1 + vv_size) + 24*fpitch B).
// ---------------------------------------------------------------------------
template <bool SORT>
__global__ __launch_bounds__(256) void ref_step_kernel(
    const double2 *__restrict__ fn_src, double2 *__restrict__ fn_dst, const uint *__restrict__ fv,
    const double *__restrict__ fv_weights, uint fv_pitch, uint fvw_pitch, const uint *__restrict__ nbr,
    const double *__restrict__ vtxW, uint vw_pitch, const double *__restrict__ vertex_weights,
    uint vv_pitch, uint vws_pitch, uint vv_size, const float4 *__restrict__ grads, uint he_pitch,
    const uint *__restrict__ vertex_parts, const uint *__restrict__ block_face_count, double dt,
    uint max_size, ModelD m) {
  extern __shared__ double s_mem[];
  // bijective XCD remap: workgroups are dealt round-robin over the 8 XCDs, so blocks b, b+8, ...
  // (one XCD, one L2) take consecutive partitions -- neighbours in space, whose gathers of each
  // other's state then hit that L2
  const uint nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7u, xcd = blockIdx.x & 7u;
  const uint p = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const uint base = vertex_parts[p];
  const uint size = vertex_parts[p + 1] - base;
  const uint nfaces = block_face_count[p];
  // A vertex has at most vv_size (= the largest valence, main.cpp:261-287) incident faces, so its
  // corner list gets a fixed slot range: no counting pass, no scan.
  uint *s_cur = reinterpret_cast<uint *>(s_mem);          // [max_size]            corners filled so far
  uint *s_ent = s_cur + max_size;                         // [max_size * vv_size]  face slot << 2 | corner
  float *s_fg = reinterpret_cast<float *>(s_ent + (size_t)max_size * vv_size);   // [6 * fv_pitch]
  const uint tid = threadIdx.x;

  for (uint i = tid; i < size; i += blockDim.x) s_cur[i] = 0;
  __syncthreads();

  /* face gradients (FEM_common.cu:99-116) parked in LDS; each owned corner appended to its vertex's list */
  const size_t fvb = (size_t)p * 3 * fv_pitch, fwb = (size_t)p * 3 * fvw_pitch, heb = (size_t)p * 2 * he_pitch;
  for (uint t = tid; t < nfaces; t += blockDim.x) {
    const uint i0 = fv[fvb + t], i1 = fv[fvb + fv_pitch + t], i2 = fv[fvb + 2 * (size_t)fv_pitch + t];
    const double2 v1 = fn_src[i2], a = fn_src[i0], b = fn_src[i1];
    const float4 g12 = grads[heb + t], g13 = grads[heb + he_pitch + t];
    const float d12x = (float)(a.x - v1.x), d12y = (float)(a.y - v1.y);
    const float d13x = (float)(b.x - v1.x), d13y = (float)(b.y - v1.y);
    float *fg = s_fg + 6 * (size_t)t;
    fg[0] = g12.x * d12x + g13.x * d13x; fg[1] = g12.y * d12x + g13.y * d13x; fg[2] = g12.z * d12x + g13.z * d13x;
    fg[3] = g12.x * d12y + g13.x * d13y; fg[4] = g12.y * d12y + g13.y * d13y; fg[5] = g12.z * d12y + g13.z * d13y;
    const uint idx[3] = {i0, i1, i2};
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const uint li = idx[j] - base;                // unsigned wrap makes "below base" large
      if (li < size) {
        const uint pos = atomicAdd(&s_cur[li], 1u);
        if (pos < vv_size) s_ent[(size_t)li * vv_size + pos] = (t << 2) | (uint)j;
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
    // neighbours in groups of 8 with every load of a group in flight before the first use (padding
    // slots hold index 0 / weight 0, main.cpp:261-287: loaded, not accumulated); same summation
    // order as the reference's loop (FEM_common.cu:146-150)
    // (the first group does not wait for `end`: its loads are clamped, its sums predicated)
    for (uint j0 = 0; j0 < (vv_size ? max(end, 1u) : 0u); j0 += 8) {
      uint nIdx[8];
      double hW[8];
      double2 f[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint j = min(j0 + u, vv_size - 1);
        nIdx[u] = nbr[nb + (size_t)vv_pitch * (j + 1) + t];
        hW[u] = vertex_weights[wb + (size_t)vws_pitch * j + t];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) f[u] = fn_src[nIdx[u]];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (j0 + u < end) { lx += f[u].x * hW[u]; ly += f[u].y * hW[u]; }
    }
    // this vertex's corners in ascending (face slot, corner) order: weighted gradient sums in float,
    // weight sum in double (FEM_common.cu:118-132), then the normalised dot product (:153-158)
    const uint eb = t * vv_size, ee = eb + min(s_cur[t], vv_size);
    for (uint i = eb + 1; SORT && i < ee; i++) {      // insertion sort, a handful of entries
      const uint key = s_ent[i];
      uint k = i;
      while (k > eb && s_ent[k - 1] > key) { s_ent[k] = s_ent[k - 1]; k--; }
      s_ent[k] = key;
    }
    float nvx = 0, nvy = 0, nvz = 0, cvx = 0, cvy = 0, cvz = 0;
    double wg = 0;
    for (uint i = eb; i < ee; i++) {
      const uint ent = s_ent[i], slot = ent >> 2, j = ent & 3u;
      const double weight = fv_weights[fwb + (size_t)j * fvw_pitch + slot];
      const float *fg = s_fg + 6 * (size_t)slot;
      nvx += (float)(fg[0] * weight); nvy += (float)(fg[1] * weight); nvz += (float)(fg[2] * weight);
      cvx += (float)(fg[3] * weight); cvy += (float)(fg[4] * weight); cvz += (float)(fg[5] * weight);
      wg += weight;
    }
    double dotP = 0;
    if (wg > 0) {
      const float w = (float)wg;
      dotP = (nvx / w) * (cvx / w) + (nvy / w) * (cvy / w) + (nvz / w) * (cvz / w);
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
// PK (fp32 plans whose operator comes from the host assembly): the operator stream in 12-byte records -- the weight
// in fp32 and the three gradient coefficients as 21-bit integers against the row's scale, as in the fused kernel below
// (there: the format and its accuracy) -- at 3 * e words of `coef`, with rscale[v] the scale of row v: 100 B per
// vertex-step (valence 6) instead of 120 of operator.
template <typename T, bool NT, bool IMEX = false, bool PK = false>
__global__ __launch_bounds__(256) void sell_step_kernel(
    const typename Vec<T>::t2 *__restrict__ src, typename Vec<T>::t2 *__restrict__ dst,
    const uint *__restrict__ slice_off, const uint *__restrict__ idx,
    const typename Vec<T>::t4 *__restrict__ coef, uint V, uint nslices, uint nblocks, ModelT<T> m,
    const float *__restrict__ rscale = nullptr) {
  static_assert(!PK || sizeof(T) == 4, "packed operator records: fp32 plans");
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
  float rs = 0;
  if constexpr (PK) rs = NT ? __builtin_nontemporal_load(&rscale[vs]) : rscale[vs];
  auto entry = [&](uint e) {
    const uint j = NT ? __builtin_nontemporal_load(&idx[e]) : idx[e];     // read-once streams: do not evict the state
    T4 c;
    if constexpr (PK) {
      const uint *rec = reinterpret_cast<const uint *>(coef) + 3 * (size_t)e;
      const uint w0 = NT ? __builtin_nontemporal_load(rec) : rec[0], a = NT ? __builtin_nontemporal_load(rec + 1) : rec[1],
                 b2 = NT ? __builtin_nontemporal_load(rec + 2) : rec[2];
      c.x = __uint_as_float(w0);
      c.y = (T)((float)((int)(a << 11) >> 11) * rs);
      c.z = (T)((float)((int)(__builtin_amdgcn_alignbit(b2, a, 21) << 11) >> 11) * rs);
      c.w = (T)((float)((int)(b2 << 1) >> 11) * rs);
    } else {
      c = NT ? ntload(&coef[e]) : coef[e];
    }
    const T2 f = src[j];
    const T dn = f.x - s.x, dc = f.y - s.y;
    lapn += c.x * dn; lapc += c.x * dc;
    gnx += c.y * dn; gny += c.z * dn; gnz += c.w * dn;
    gcx += c.y * dc; gcy += c.z * dc; gcz += c.w * dc;
  };
  // six entries (the common valence) at a time with all their loads in flight before the first use;
  // the entries of a wider slice beyond a multiple of six two at a time
  uint e = off + lane;
#pragma unroll 1
  for (; e + 5 * 64 < end; e += 6 * 64) {
#pragma unroll
    for (uint u = 0; u < 6; u++) entry(e + 64 * u);
  }
#pragma unroll 2
  for (; e < end; e += 64) entry(e);
  const T2 o = IMEX ? imex_rhs_update<T>(s, lapc, gnx, gny, gnz, gcx, gcy, gcz, m)
                    : reaction_update<T>(s, lapn, lapc, gnx, gny, gnz, gcx, gcy, gcz, m);
  if (v < V) dst[v] = o;
}

// ---------------------------------------------------------------------------
// Fused k-step kernel.  One workgroup per tile (pack.cpp:packTileLayout): the tile's owned rows
// plus k rings of neighbours are staged in LDS (two ping-pong buffers of m[k] pairs), the
// operator rows of the first m[k-1] local rows are loaded ONCE into registers (thread t holds rows
// t, t+TPB, ...: R rows x W entries x (float4 coef + 16-bit local column id); default R = 3), and the workgroup
// takes up to k time steps before writing its owned rows back: step s recomputes rows
// [0, m[k-1-s]), whose neighbours [0, m[k-s]) are valid from the step before.  HBM traffic per
// vertex-step drops from 136 B to about (ring ratio * 108 + ghost state/ids + 8)/k B: measured 45 B at
// k = 5 (530-row partitions, R = 2) and 38.5 B at k = 6 (760-row partitions, R = 3) on the 16M sphere.
// ---------------------------------------------------------------------------
// packed column offsets of the register rows: in VGPRs (R * W/2 of them), or parked in LDS and re-read
// every step.  LDS was the way to stay under 128 VGPRs while the compiler kept rearranged copies of all
// coefficients; with the accumulators paired along the records (below) the registers are there, and the
// LDS round trip in front of every row's gathers is not (+1 %).
constexpr bool tileOffsetsInLds(int rowsPerThread) { return rowsPerThread >= 4; }   // 4 rows need the 12 VGPRs back
#ifndef FEM_TILE_ROW_FENCE
#define FEM_TILE_ROW_FENCE 1
#endif
constexpr bool kTileRowFence = FEM_TILE_ROW_FENCE != 0;

// PK = 1 (fp32 plans whose operator comes from the host assembly, FEM_TILE_PACKED): the operator stream in 12-byte
// records instead of 16 -- the Laplacian weight as it is (fp32) and the three gradient coefficients as 21-bit signed
// integers against a per-row fp32 scale (row maximum / (2^20 - 1): a rounding error of at most 4.8e-7 of the row's
// largest coefficient, 8 x an fp32 rounding), decoded once per launch into the same register rows.  A tile's block
// keeps its place and size in the coefficient array (16 bytes per entry from row_off on): the W * rows_pad records
// fill its first three quarters, the rows_pad scales follow; the extras of wide rows stay four floats behind the block.
// State accesses of the tile body.  COH (the persistent kernel, where another workgroup of the SAME launch wrote the rows):
// agent-scope relaxed atomics = write-through stores / loads that bypass the L1 (global_store/load ... sc1), 8 bytes each.
template <bool COH, typename T2> __device__ __forceinline__ T2 state_load(const T2 *p) {
  if constexpr (COH && sizeof(T2) == 8) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    T2 o;
    __builtin_memcpy(&o, &v, 8);
    return o;
  } else {
    return *p;
  }
}
template <bool COH, typename T2> __device__ __forceinline__ void state_store(T2 *p, T2 v) {
  if constexpr (COH && sizeof(T2) == 8) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}

// MODE 1 (tile_rhs_kernel, semi-implicit stepping on fused tiles): the step's update is the explicit right-hand side
// b = u + dt (reaction + chemotaxis terms) without the diffusion terms and without the clamp (imex_rhs_update).
// MODE 2 (tile_rhs_cheb_kernel): that right-hand side, then -- in the same launch, the operator rows still in registers --
// the first tail->nit <= k - 1 Chebyshev iterations of the two diffusion solves from x_0 = b (kernels_implicit.hpp has
// the iteration): the right-hand side used up one ring of the tile, every iteration uses up one more.  Writes b (dst),
// the iterate and the direction of the owned rows.
template <typename T> struct ChebArgs { T c1n[8], c2n[8], c1c[8], c2c[8], dtDn, dtDc; int first; };
template <typename T> struct ChebTail { typename Vec<T>::t2 *xout, *dout; int nit; ChebArgs<T> a; };

template <typename T, int TPB, int R, int W, int X, int PRO, int PK, int COH, int MODE = 0>
__device__ __forceinline__ void tile_body(
    const typename Vec<T>::t2 *__restrict__ src, typename Vec<T>::t2 *__restrict__ dst,
    const uint *__restrict__ meta, const uint *__restrict__ ghost, const uint *__restrict__ idx2,
    const typename Vec<T>::t4 *__restrict__ coef, const uint *__restrict__ xinfo, uint tile,
    uint maxLocal, uint maxExtra, int nsteps, const ModelT<T> &m, const ChebTail<T> *tail = nullptr) {
  static_assert(PK != 1 || sizeof(T) == 4, "packed operator records: fp32 plans");
  static_assert(PK != 2 || sizeof(T) == 8, "mixed operator records: fp64 plans");
  typedef typename Vec<T>::t2 T2;
  typedef typename Vec<T>::t4 T4;
  extern __shared__ __align__(16) unsigned char tile_smem[];
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
  constexpr bool IXL = tileOffsetsInLds(R);
  const uint xoff = mt[14], xcnt = X ? mt[15] : 0u;      // extras of wide rows in this tile (X = 0: never any)
  const uint t = threadIdx.x;

  // tile-uniform (scalar) bases + 32-bit lane offsets keep the address arithmetic out of VGPRs
  const T4 *__restrict__ tcoef = coef + roff;
  const uint *__restrict__ tpk = reinterpret_cast<const uint *>(tcoef);                        // PK: 12-byte records ...
  const float *__restrict__ trs = reinterpret_cast<const float *>(tpk + 3 * (size_t)(W * rpad));   // ... then the row scales
  const uint *__restrict__ tidx = idx2 + (roff >> 1);
  // Every load below is unconditional with a clamped index (no branch => no wait at a branch
  // merge), so the whole prologue is in flight at once: R*(W + W/2) operator loads, then the
  // ghost ids, then the state gathers; the first wait is the one in front of the LDS writes.
  T4 cf[R][W];
  uint ix[R][W / 2];
  uint pk[PK == 1 ? R : 1][PK == 1 ? W : 1][3];
  float rs[PK == 1 ? R : 1];
  double mw[PK == 2 ? R : 1][PK == 2 ? W : 1];        // PK == 2: Laplacian weights in fp64 ...
  float mg[PK == 2 ? R : 1][PK == 2 ? W : 1][3];      // ... gradient coefficients in fp32
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
  // Wide rows' extras (entries beyond W) are staged through registers into LDS.  X == 2 (plans that
  // have any): first of all, where the staging waits for its own loads only; after the operator
  // loads it makes every wave wait for those before the state gathers can leave (scan8m 165 ->
  // 170 G).  X == 1 (plans without extras, the block never runs): kept at the old place, because
  // that build schedules the regular path 4 % faster than either alternative.
  if (X == 2 && xcnt) {
    // all loads of the staging in flight before the first LDS write (clamped indices, predicated
    // writes): rpad <= TPB * R rows, and the extras in passes of 2 * TPB
    uint xi[R];
#pragma unroll
    for (int q = 0; q < R; q++) xi[q] = xinfo[xoff + min(t + TPB * q, rpad - 1)];      // first extra | count << 16 per row
    for (uint i0 = 0; i0 < xcnt; i0 += 2 * TPB) {
      T4 xc[2];
      uint xo[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint i = min(i0 + t + TPB * u, xcnt - 1);
        xc[u] = tcoef[W * rpad + i];
        xo[u] = xinfo[xoff + rpad + i];
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint i = i0 + t + TPB * u;
        if (i < xcnt) { xs_coef[i] = xc[u]; xs_off[i] = xo[u] * (uint)sizeof(T2); }
      }
    }
#pragma unroll
    for (int q = 0; q < R; q++)
      if (t + TPB * q < rpad) xs_info[t + TPB * q] = xi[q];
  }
  if (PRO >= 1) loadGhostIds(0);
  // X == 2: pin the id loads here.  Left to itself that build sinks them below the operator loads, next to the state
  // gathers that use them, so the gathers wait behind the whole operator stream (s_waitcnt vmcnt(0) in front of them
  // instead of vmcnt(32) in the X == 1 build): the "scheduling accident" that cost wide-row plans ~20 %.
#pragma unroll
  for (int q = 0; q < R; q++) {
    const uint r = min(t + TPB * q, rpad - 1);         // rows in [rows, rpad) are zero padding
#pragma unroll
    for (int w = 0; w < W / 2; w++)                      // ids < 4096: both halves scale
      ix[q][w] = (PRO == 3 ? __builtin_nontemporal_load(&tidx[w * rpad + r]) : tidx[w * rpad + r]) * (uint)sizeof(T2);
  }
  if constexpr (PK == 2) {
    // the tile's block: W * rows_pad weights (fp64), then W * rows_pad gradient coefficients of three floats
    const double *__restrict__ tw = reinterpret_cast<const double *>(tcoef);
    const float *__restrict__ tg = reinterpret_cast<const float *>(tw + (size_t)W * rpad);
#pragma unroll
    for (int q = 0; q < R; q++) {
      const uint r = min(t + TPB * q, rpad - 1);
#pragma unroll
      for (int w = 0; w < W; w++) {
        const double *wp = tw + (w * rpad + r);
        const float *gp = tg + 3 * (size_t)(w * rpad + r);
        mw[q][w] = PRO >= 2 ? __builtin_nontemporal_load(wp) : *wp;
#pragma unroll
        for (int d = 0; d < 3; d++) mg[q][w][d] = PRO >= 2 ? __builtin_nontemporal_load(gp + d) : gp[d];
      }
    }
  } else if constexpr (PK == 1) {
    // scalar base of every entry slot + one 32-bit byte offset per row: the loads take the (SGPR base, VGPR offset)
    // form, no 64-bit address pair per load
#pragma unroll
    for (int q = 0; q < R; q++) {
      const uint boff = min(t + TPB * q, rpad - 1) * 12u;
#pragma unroll
      for (int w = 0; w < W; w++) {
        const unsigned char *slot = reinterpret_cast<const unsigned char *>(tpk + 3 * (size_t)(w * rpad));
        const uint *rec = reinterpret_cast<const uint *>(slot + boff);
#pragma unroll
        for (int d = 0; d < 3; d++) pk[q][w][d] = PRO >= 2 ? __builtin_nontemporal_load(rec + d) : rec[d];
      }
    }
#pragma unroll
    for (int q = 0; q < R; q++) {
      const float *rp = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(trs) + min(t + TPB * q, rpad - 1) * 4u);
      rs[q] = PRO >= 2 ? __builtin_nontemporal_load(rp) : *rp;
    }
  } else {
#pragma unroll
    for (int q = 0; q < R; q++) {
      const uint r = min(t + TPB * q, rpad - 1);
#pragma unroll
      for (int w = 0; w < W; w++) cf[q][w] = PRO >= 2 ? ntload(&tcoef[w * rpad + r]) : tcoef[w * rpad + r];
    }
  }
  if (X == 1 && xcnt) {                                  // tile-uniform; see X == 2 above
    for (uint i = t; i < rpad; i += TPB) xs_info[i] = xinfo[xoff + i];      // first extra | count << 16 per row
    for (uint i = t; i < xcnt; i += TPB) {
      xs_coef[i] = tcoef[W * rpad + i];
      xs_off[i] = xinfo[xoff + rpad + i] * (uint)sizeof(T2);
    }
  }
  // (an IR pass moves them past the guard of the staging loop below, the only place that uses them: an opaque use in
  //  front of that guard, behind the operator loads, keeps them where they were written -- first)
  if constexpr (X == 2 && PRO >= 1) {
#pragma unroll
    for (int u = 0; u < SU; u++) asm volatile("" : "+v"(gi[u]));
  }
  for (uint i0 = 0; i0 < mk; i0 += SU * TPB) {
    T2 sv[SU];
    if (PRO < 1 || i0 > 0) loadGhostIds(i0);
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const uint i = i0 + t + TPB * u;
      sv[u] = state_load<(COH == 1 || COH == 2)>(&src[i < n0 ? base + i : gi[u]]);
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
  if constexpr (PK == 1) {
    // decode the records into the register rows (all loads have landed: the state writes above waited for the
    // gathers, which were issued last)
#pragma unroll
    for (int q = 0; q < R; q++)
#pragma unroll
      for (int w = 0; w < W; w++) {
        const uint a = pk[q][w][1], b2 = pk[q][w][2];
        const int qx = (int)(a << 11) >> 11;                                        // bits  0..20
        const int qy = (int)(__builtin_amdgcn_alignbit(b2, a, 21) << 11) >> 11;      // bits 21..41
        const int qz = (int)(b2 << 1) >> 11;                                        // bits 42..62
        cf[q][w].x = __uint_as_float(pk[q][w][0]);
        cf[q][w].y = (T)((float)qx * rs[q]); cf[q][w].z = (T)((float)qy * rs[q]); cf[q][w].w = (T)((float)qz * rs[q]);
      }
  }
  __syncthreads();

  T2 bb[MODE == 2 ? R : 1];                              // MODE 2: the right-hand side of this thread's rows
  for (int s = 0; s < nsteps; s++) {
    const uint lim = mt[5 + k - 1 - s];
#pragma unroll
    for (int q = 0; q < R; q++) {
      const uint r = t + TPB * q;
      if (r < lim) {
        const T2 self = cur[r];
        // accumulators paired the way a coefficient record lies in its registers: (w, Cx) and (Cy, Cz)
        // times dn, the same times dc -- four packed FMAs per entry whose coefficient operands are the
        // loaded register pairs themselves (any other pairing makes the compiler keep rearranged copies
        // of all coefficients: + 43 VGPRs)
        typedef T V2 __attribute__((ext_vector_type(2)));
        V2 an = {0, 0}, bn = {0, 0}, ac = {0, 0}, bc = {0, 0};       // (lap, gx), (gy, gz) of n and of c
        // (the first row's offsets stay in registers where that costs no spill -- builds without the wide-row
        //  staging: one LDS round trip less per step, +0.6 % in a same-box A/B; a second row's spill in the loop: -11 %)
        if (IXL && q >= (X == 1 ? 1 : 0)) {
#pragma unroll
          for (int w = 0; w < W / 2; w++) ix[q][w] = xs_ix[w * (TPB * R) + r];
        }
#pragma unroll
        for (int w = 0; w < W / 2; w++) asm volatile("" : "+v"(ix[q][w]));   // keep the unpacking in the loop, not in VGPRs
        // PK == 2 (the reference's own mix, FEM_common.cu:99-150: double2 state and Laplacian, float4 gradient vectors
        // and float3 accumulators): the differences and the Laplacian sums in fp64, the differences narrowed to fp32 for
        // the gradient sums, which run in fp32 -- (gx, gy) of n, (gx, gy) of c and (gz of n, gz of c) as packed pairs
        typedef float F2 __attribute__((ext_vector_type(2)));
        F2 gn01 = {0, 0}, gc01 = {0, 0}, g2 = {0, 0};
        T mlapn = 0, mlapc = 0;
#pragma unroll
        for (int w = 0; w < W; w++) {
          const uint joff = (w & 1) ? (ix[q][w >> 1] >> 16) : (ix[q][w >> 1] & 0xffffu);   // byte offset in the buffer
          const T2 f = *reinterpret_cast<const T2 *>(reinterpret_cast<const unsigned char *>(cur) + joff);
          const T dn = f.x - self.x, dc = f.y - self.y;
          if constexpr (PK == 2) {
            mlapn += mw[q][w] * dn; mlapc += mw[q][w] * dc;
            const F2 cxy = {mg[q][w][0], mg[q][w][1]}, d2 = {(float)dn, (float)dc};
            gn01 += cxy * d2.x; gc01 += cxy * d2.y; g2 += mg[q][w][2] * d2;
          } else {
            const T4 c = cf[q][w];
            const V2 cxy = {c.x, c.y}, czw = {c.z, c.w};
            an += cxy * dn; bn += czw * dn;
            ac += cxy * dc; bc += czw * dc;
          }
        }
        T lapn = an.x, gnx = an.y, gny = bn.x, gnz = bn.y, lapc = ac.x, gcx = ac.y, gcy = bc.x, gcz = bc.y;
        if constexpr (PK == 2) {
          lapn = mlapn; lapc = mlapc;
          gnx = gn01.x; gny = gn01.y; gnz = g2.x; gcx = gc01.x; gcy = gc01.y; gcz = g2.y;
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
        if constexpr (MODE == 2) nxt[r] = bb[q] = imex_rhs_update<T>(self, lapc, gnx, gny, gnz, gcx, gcy, gcz, m);
        else if constexpr (MODE == 1) nxt[r] = imex_rhs_update<T>(self, lapc, gnx, gny, gnz, gcx, gcy, gcz, m);
        else nxt[r] = reaction_update<T>(self, lapn, lapc, gnx, gny, gnz, gcx, gcy, gcz, m);
      }
      if constexpr (kTileRowFence) __builtin_amdgcn_sched_barrier(0);      // one row at a time: keeps the live set within 128 VGPRs
    }
    __syncthreads();
    T2 *tmp = cur; cur = nxt; nxt = tmp;
  }
  if constexpr (MODE == 2) {
    // cur = b = x_0 on the rows of all but the outermost ring.  Iteration s reads one ring more than it writes.
    const ChebArgs<T> &a = tail->a;
    T2 dd[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
      dd[q].x = 0; dd[q].y = 0;
      if (t + TPB * q < n0) dst[base + t + TPB * q] = bb[q];
    }
    for (int s = 0; s < tail->nit; s++) {
      const uint lim = mt[5 + k - 2 - s];
      const T c1n = a.c1n[s], c2n = a.c2n[s], c1c = a.c1c[s], c2c = a.c2c[s];
#pragma unroll
      for (int q = 0; q < R; q++) {
        const uint r = t + TPB * q;
        if (r < lim) {
          const T2 self = cur[r];
          if (IXL && q >= (X == 1 ? 1 : 0)) {
#pragma unroll
            for (int w = 0; w < W / 2; w++) ix[q][w] = xs_ix[w * (TPB * R) + r];
          }
#pragma unroll
          for (int w = 0; w < W / 2; w++) asm volatile("" : "+v"(ix[q][w]));
          T ln = 0, lc = 0;
#pragma unroll
          for (int w = 0; w < W; w++) {
            const uint joff = (w & 1) ? (ix[q][w >> 1] >> 16) : (ix[q][w >> 1] & 0xffffu);
            const T2 f = *reinterpret_cast<const T2 *>(reinterpret_cast<const unsigned char *>(cur) + joff);
            T wt;
            if constexpr (PK == 2) wt = mw[q][w]; else wt = cf[q][w].x;
            ln += wt * (f.x - self.x); lc += wt * (f.y - self.y);
          }
          if (X && xcnt) {
            const uint xi = xs_info[r];
            for (uint e = xi & 0xffffu, ee = e + (xi >> 16); e < ee; e++) {
              const T2 f = *reinterpret_cast<const T2 *>(reinterpret_cast<const unsigned char *>(cur) + xs_off[e]);
              const T wt = xs_coef[e].x;
              ln += wt * (f.x - self.x); lc += wt * (f.y - self.y);
            }
          }
          const T rn = bb[q].x - (self.x - a.dtDn * ln), rc = bb[q].y - (self.y - a.dtDc * lc);
          dd[q].x = c1n * dd[q].x + c2n * rn; dd[q].y = c1c * dd[q].y + c2c * rc;
          T2 o;
          o.x = self.x + dd[q].x; o.y = self.y + dd[q].y;
          nxt[r] = o;
        }
      }
      __syncthreads();
      T2 *tmp = cur; cur = nxt; nxt = tmp;
    }
    for (uint i = t; i < n0; i += TPB) tail->xout[base + i] = cur[i];
#pragma unroll
    for (int q = 0; q < R; q++)
      if (t + TPB * q < n0) tail->dout[base + t + TPB * q] = dd[q];
    return;
  }
  for (uint i = t; i < n0; i += TPB) state_store<(COH == 1 || COH == 3)>(&dst[base + i], cur[i]);
}

template <typename T, int TPB, int R, int W, int MINW, int X, int PRO, int PK = 0>
__global__ __launch_bounds__(TPB, MINW) void tile_step_kernel(
    const typename Vec<T>::t2 *__restrict__ src, typename Vec<T>::t2 *__restrict__ dst,
    const uint *__restrict__ meta, const uint *__restrict__ ghost, const uint *__restrict__ idx2,
    const typename Vec<T>::t4 *__restrict__ coef, const uint *__restrict__ xinfo, uint tile0, uint ntiles,
    uint maxLocal, uint maxExtra, int nsteps, ModelT<T> m) {
  const uint b = blockIdx.x, q8 = ntiles >> 3, r8 = ntiles & 7u, x = b & 7u;
  const uint tile = tile0 + (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (b >> 3);   // tiles [tile0, tile0 + ntiles)
  tile_body<T, TPB, R, W, X, PRO, PK, 0>(src, dst, meta, ghost, idx2, coef, xinfo, tile, maxLocal, maxExtra, nsteps, m);
}

// one "step" of every tile with the explicit right-hand side of a semi-implicit step as its update (kernels_implicit.hpp)
template <typename T, int TPB, int R, int W, int MINW, int X, int PK>
__global__ __launch_bounds__(TPB, MINW) void tile_rhs_kernel(
    const typename Vec<T>::t2 *__restrict__ src, typename Vec<T>::t2 *__restrict__ dst,
    const uint *__restrict__ meta, const uint *__restrict__ ghost, const uint *__restrict__ idx2,
    const typename Vec<T>::t4 *__restrict__ coef, const uint *__restrict__ xinfo, uint ntiles,
    uint maxLocal, uint maxExtra, ModelT<T> m) {
  const uint b = blockIdx.x, q8 = ntiles >> 3, r8 = ntiles & 7u, x = b & 7u;
  const uint tile = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (b >> 3);
  tile_body<T, TPB, R, W, X, 1, PK, 0, 1>(src, dst, meta, ghost, idx2, coef, xinfo, tile, maxLocal, maxExtra, 1, m);
}

// the same right-hand side and, with the operator rows still in registers, the first nit <= k - 1 Chebyshev iterations
template <typename T, int TPB, int R, int W, int MINW, int X, int PK>
__global__ __launch_bounds__(TPB, MINW) void tile_rhs_cheb_kernel(
    const typename Vec<T>::t2 *__restrict__ src, typename Vec<T>::t2 *__restrict__ bout,
    const uint *__restrict__ meta, const uint *__restrict__ ghost, const uint *__restrict__ idx2,
    const typename Vec<T>::t4 *__restrict__ coef, const uint *__restrict__ xinfo, uint ntiles,
    uint maxLocal, uint maxExtra, ModelT<T> m, ChebTail<T> tail) {
  const uint b = blockIdx.x, q8 = ntiles >> 3, r8 = ntiles & 7u, x = b & 7u;
  const uint tile = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (b >> 3);
  tile_body<T, TPB, R, W, X, 1, PK, 0, 2>(src, bout, meta, ghost, idx2, coef, xinfo, tile, maxLocal, maxExtra, 1, m, &tail);
}

// ---------------------------------------------------------------------------
// The same tiles without a launch-wide barrier between groups of k steps (chosen by tile count,
// device.hip:persistRule).  A fixed grid of workgroups draws (group, tile) items from tickets, in group-major tile order.
// Item (g, T) reads buffer g % 2 and writes the other; it may start when every tile within T's rings (dep[], symmetric,
// T itself included) has PUBLISHED group g - 1: flag[tile] = number of groups that tile has finished.  The same condition
// covers the write-after-read hazard on the ping-pong buffers: a neighbour one group ahead would need T's flag first.
// Progress: a workgroup holds an item only while it runs or waits, and an item waits only for items of the group before,
// which have smaller tickets in their ranges; workgroups start in launch order, so each of the eight ranges has a live
// workgroup from the moment eight are resident until the range is exhausted (tests/test_persist_protocol.py models it).
// Hand-off form (MI355X guide, Guideline 16, R1): every state store write-through (sc1), every storing wave drains
// (s_waitcnt vmcnt(0)), the workgroup's barrier, then the flag store (sc1); consumer: one wave polls the flags (sc1
// loads, bounded: a give-up sets ctl[128] and every workgroup leaves), barrier, then every state load sc1.
// CM (rate probes, FEM_TILE_PERSIST_COH): 1 as described; 0 plain loads and stores, 2 sc1 loads only, 3 sc1 stores only
// -- all four run at the same rate (200 G DOF-it/s with one ticket for all tiles, 304..336 G with the eight ranges,
// 16.4M rows): what the kernel is sensitive to is which XCD's L2 a tile's neighbours went through, not the sc1 bits.
// ---------------------------------------------------------------------------
template <typename T, int TPB, int R, int W, int MINW, int X, int PRO, int PK, int CM = 1>
__global__ __launch_bounds__(TPB, MINW) void tile_step_persist_kernel(
    typename Vec<T>::t2 *__restrict__ buf0, typename Vec<T>::t2 *__restrict__ buf1,
    const uint *__restrict__ meta, const uint *__restrict__ ghost, const uint *__restrict__ idx2,
    const typename Vec<T>::t4 *__restrict__ coef, const uint *__restrict__ xinfo, uint tile0, uint ntiles,
    uint maxLocal, uint maxExtra, int ksteps, int lastSteps, uint ngroups, ModelT<T> m, uint *__restrict__ ctl,
    uint *__restrict__ flags, const uint *__restrict__ depOff, const uint *__restrict__ dep, int useDeps, uint ldsItem) {
  extern __shared__ __align__(16) unsigned char tile_smem[];
  uint *itemSlot = reinterpret_cast<uint *>(tile_smem + ldsItem);      // [0] item, [1] give-up word at the draw, [2] after the poll
  const uint t = threadIdx.x;
  // Tiles in eight contiguous ranges, one per XCD, each with a ticket of its own (ctl[16 * x]; the give-up word is
  // ctl[128]): workgroup b runs on XCD b % 8 under dispatch-order placement, so neighbouring tiles share an L2 as in the
  // launch-per-group kernel's blockIdx -> tile map.  Placement is a matter of speed only: every hand-off is agent scope.
  const uint xcd = blockIdx.x & 7u, q8 = ntiles >> 3, r8 = ntiles & 7u;
  const uint xlo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, xn = q8 + (xcd < r8 ? 1u : 0u);
  const uint total = ngroups * xn;
  uint *__restrict__ ticket = ctl + 16 * xcd;
  uint *__restrict__ giveup = ctl + 128;
  // Everything one thread would do for its workgroup (draw a ticket, poll, publish) is done by WAVE 0 WITH ALL ITS LANES
  // behind a wave-uniform branch.  An `if (threadIdx.x == 0)` in this loop is lane-divergent control flow whose result
  // feeds the loop's exit: the structuriser then runs the ticket draw in an outer loop that lane 0 only reaches once
  // every other lane has left the inner one -- which they never do, since they wait for the ticket (the first version
  // of this kernel span on its first item for ever; most likely round 2's unresolved "multi-tile bug" too).
  const bool wave0 = __builtin_amdgcn_readfirstlane(t >> 6) == 0;
  for (;;) {
    if (wave0) {
      // 64 lanes add 1 each (the compiler folds them into one add of 64 and hands out consecutive values): the item is
      // the first lane's value / 64; every lane writes the same two words.  (Drawing the NEXT item's ticket here, to
      // hide the round trip behind the running item: no gain at 1M and 16M rows, -19 % at 256k, where a workgroup
      // then sits on an item that an idle one could run.)
      const uint v = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      itemSlot[0] = __builtin_amdgcn_readfirstlane(v) >> 6;
      itemSlot[1] = __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const uint item = __builtin_amdgcn_readfirstlane(itemSlot[0]), aborted = __builtin_amdgcn_readfirstlane(itemSlot[1]);
    if (item >= total || aborted) break;
    const uint g = item / xn, tile = tile0 + xlo + (item - g * xn);
    uint stop = 0;
    if (useDeps && g > 0) {
      if (wave0) {
        const uint lane = t & 63u;
        const uint d0 = depOff[tile], nd = depOff[tile + 1] - d0;
        for (uint d = lane; d < nd; d += 64) {
          const uint other = dep[d0 + d];
          uint spins = 0;
          while (__hip_atomic_load(&flags[other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g) {
            __builtin_amdgcn_s_sleep(4);
            // bounded: a give-up (or anybody else's) ends the whole grid within microseconds
            if (++spins > (1u << 16)) { __hip_atomic_store(giveup, 1u + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if ((spins & 63u) == 0 && __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the state loads below the poll
        itemSlot[2] = __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a word of its own: slow waves may still be reading [1])
      }
      __syncthreads();
      stop = __builtin_amdgcn_readfirstlane(itemSlot[2]);
    }
    if (stop) break;                                               // somebody gave up: leave without touching the state
    typename Vec<T>::t2 *src = (g & 1u) ? buf1 : buf0, *dst = (g & 1u) ? buf0 : buf1;
    tile_body<T, TPB, R, W, X, PRO, PK, CM>(src, dst, meta, ghost, idx2, coef, xinfo, tile, maxLocal, maxExtra,
                                            g + 1 == ngroups ? lastSteps : ksteps, m);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every storing wave drains its write-through stores
    __syncthreads();
    if (wave0) __hip_atomic_store(&flags[tile], g + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all lanes, one word
  }
}


