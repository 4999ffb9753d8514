// kernels_assembly.hpp -- operator assembly on the device (initMeshStats on the GPU), per-face records, deformation
// Included by device.hip inside its anonymous namespace (one translation unit: the kernels are
// launched from the C-ABI functions there).
#pragma once

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
// Terms of entry (v -> j) from the edge vectors ej = j - v, e1 = o1 - v, e2 = o2 - v (o1 / o2 = the
// vertices opposite the edge in the faces left of v->j / of j->v; has1 / has2: that face exists).
template <typename T>
__device__ __forceinline__ typename Vec<T>::t4 entry_terms(const T *ej, const T *e1, const T *e2, bool real, bool has1, bool has2,
                                                           T &area3, T &theta, T &maxe2) {
  typename Vec<T>::t4 a;
  a.x = a.y = a.z = a.w = 0;
  if (has1) {                                          // face (v, j, o1): p = j, q = o1
    const FanFace<T> f = fan_face<T>(ej, e1);
    const T rcn = dev_rsqrt(f.c2);                     // 1 / |c|
    const T ang = dev_acos(f.d * dev_rsqrt(f.l1 * f.l2));
    a.x += (f.l2 - f.d) * rcn;                         // cot at o1
    const T s = ang * rcn * rcn;                       // angle * grad phi_j, grad phi_p = e2 x c / |c|^2
    a.y += s * (e1[1] * f.cz - e1[2] * f.cy); a.z += s * (e1[2] * f.cx - e1[0] * f.cz); a.w += s * (e1[0] * f.cy - e1[1] * f.cx);
    theta += ang; area3 += f.c2 * rcn * (T)(1.0 / 6.0);   // area / 3 = |c| / 6
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

// row normalisation: w = cot sum / (2 lumped mass), C = angle-weighted gradient sum / angle sum; the two
// reciprocals are taken once per row (a true division per component cost a fifth of the kernel)
template <typename T> struct RowScale { T inv2a, invTheta; };
template <typename T>
__device__ __forceinline__ RowScale<T> row_scale(T area3, T theta) {
  RowScale<T> r;
  r.inv2a = area3 > 0 ? (T)1 / (2 * area3) : (T)0;
  r.invTheta = theta > 0 ? (T)1 / theta : (T)0;
  return r;
}
template <typename T>
__device__ __forceinline__ typename Vec<T>::t4 entry_scale(const typename Vec<T>::t4 &a, const RowScale<T> &r) {
  typename Vec<T>::t4 o;
  o.x = a.x * r.inv2a; o.y = a.y * r.invTheta; o.z = a.z * r.invTheta; o.w = a.w * r.invTheta;
  return o;
}

// Table entry = (mesh id of the neighbour j, s1 | s2 << 8 | valid << 16): o1 and o2 are neighbours of v
// as well, named by their slot in this row, so a row gathers each neighbour position ONCE.
// W <= 12: fully unrolled, edge vectors and unnormalised terms in registers, o1/o2 picked by
// compare-select chains.  W = 0: any width, a plain loop that re-gathers o1/o2 through the table
// and parks the unnormalised terms in the output array, rescaled in a second sweep.
template <typename T, int W, int SPLIT = W>
__device__ __forceinline__ T assemble_row(const typename Vec<T>::t4 *__restrict__ pos, const uint2 *__restrict__ topo,
                                          typename Vec<T>::t4 *__restrict__ coef, uint vg, size_t first, size_t stride,
                                          uint width, size_t xfirst = 0) {
  typedef typename Vec<T>::t4 T4;
  auto at = [&](int k) { return k < SPLIT ? first + k * stride : xfirst + (size_t)(k - SPLIT); };
  const T4 P = pos[vg];
  T area3 = 0, theta = 0, maxe2 = 0;
  if constexpr (W > 0) {
    // every load unconditional (index clamped, value selected afterwards): a load inside a branch is
    // followed by a wait at the merge, which turned the 6 + 7 loads of a row into 13 round trips
    uint2 t[W];
#pragma unroll
    for (int k = 0; k < W; k++) {
      const uint2 q = topo[at((uint)k < width ? k : 0)];
      t[k].x = q.x;                                  // used by the gather in any case: keeps the load out of a branch
      t[k].y = (uint)k < width ? q.y : 0xffffu;
    }
    T e[W][3];
#pragma unroll
    for (int k = 0; k < W; k++) {
      const T4 J = pos[t[k].x];                      // padding entries name vertex 0: loaded, never used
      e[k][0] = J.x - P.x; e[k][1] = J.y - P.y; e[k][2] = J.z - P.z;
    }
    T4 acc[W];
#pragma unroll
    for (int k = 0; k < W; k++) {
      const uint s1 = t[k].y & 0xffu, s2 = (t[k].y >> 8) & 0xffu;
      const bool real = (t[k].y >> 16) & 1u;
      T e1[3] = {0, 0, 0}, e2[3] = {0, 0, 0};
#pragma unroll
      for (int u = 0; u < W; u++) {
        if (s1 == (uint)u) { e1[0] = e[u][0]; e1[1] = e[u][1]; e1[2] = e[u][2]; }
        if (s2 == (uint)u) { e2[0] = e[u][0]; e2[1] = e[u][1]; e2[2] = e[u][2]; }
      }
      acc[k] = entry_terms<T>(e[k], e1, e2, real, real && s1 < (uint)W, real && s2 < (uint)W, area3, theta, maxe2);
    }
    const RowScale<T> rs = row_scale<T>(area3, theta);
#pragma unroll
    for (int k = 0; k < W; k++)
      if ((uint)k < width) coef[at(k)] = entry_scale<T>(acc[k], rs);
  } else {
    for (int k = 0; k < (int)width; k++) {
      const uint2 tk = topo[at(k)];
      const uint s1 = tk.y & 0xffu, s2 = (tk.y >> 8) & 0xffu;
      const bool real = (tk.y >> 16) & 1u, has1 = real && s1 < width, has2 = real && s2 < width;
      const T4 J = pos[real ? tk.x : vg], O1 = pos[has1 ? topo[at((int)s1)].x : vg], O2 = pos[has2 ? topo[at((int)s2)].x : vg];
      const T ej[3] = {J.x - P.x, J.y - P.y, J.z - P.z};
      const T e1[3] = {O1.x - P.x, O1.y - P.y, O1.z - P.z};
      const T e2[3] = {O2.x - P.x, O2.y - P.y, O2.z - P.z};
      coef[at(k)] = entry_terms<T>(ej, e1, e2, real, has1, has2, area3, theta, maxe2);
    }
    const RowScale<T> rs = row_scale<T>(area3, theta);
    for (int k = 0; k < (int)width; k++) coef[at(k)] = entry_scale<T>(coef[at(k)], rs);
  }
  return dev_sqrt(maxe2);
}

// The same row when its entries are in fan order (RowOp::fanOrdered: entry k's left face ends at
// entry k + 1, from the last entry of a closed fan at entry 0): face k = (v, j_k, j_next) is computed
// ONCE and gives entry k its left-face terms and entry next its right-face terms -- half the faces of
// assemble_row and no compare-select chains.  Term by term the same expressions in the same order of
// addition (a + b = b + a exactly); the two differ only where the compiler contracts into FMAs differently.
template <typename T, int W, int SPLIT = W>
__device__ __forceinline__ T assemble_row_fan(const typename Vec<T>::t4 *__restrict__ pos, const uint2 *__restrict__ topo,
                                              typename Vec<T>::t4 *__restrict__ coef, uint vg, size_t first, size_t stride,
                                              uint width, size_t xfirst = 0) {
  typedef typename Vec<T>::t4 T4;
  auto at = [&](int k) { return k < SPLIT ? first + k * stride : xfirst + (size_t)(k - SPLIT); };
  const T4 P = pos[vg];
  T area3 = 0, theta = 0, maxe2 = 0;
  uint2 t[W];                                        // loads unconditional, as in assemble_row
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint2 q = topo[at((uint)k < width ? k : 0)];
    t[k].x = q.x;
    t[k].y = (uint)k < width ? q.y : 0xffffu;
  }
  T e[W][3];
  T4 acc[W];
#pragma unroll
  for (int k = 0; k < W; k++) {
    const T4 J = pos[t[k].x];
    e[k][0] = J.x - P.x; e[k][1] = J.y - P.y; e[k][2] = J.z - P.z;
    acc[k].x = acc[k].y = acc[k].z = acc[k].w = 0;
  }
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint s1 = t[k].y & 0xffu;
    const bool real = (t[k].y >> 16) & 1u, has1 = real && s1 < (uint)W;
    const bool wrap = (k + 1 == W) || s1 == 0u;              // the face closes the fan: its far edge is entry 0
    const int kn = k + 1 < W ? k + 1 : 0;
    T en[3];
#pragma unroll
    for (int d = 0; d < 3; d++) en[d] = wrap ? e[0][d] : e[kn][d];
    if (has1) {                                              // face (v, j_k, j_next): p = j_k, q = j_next
      const FanFace<T> f = fan_face<T>(e[k], en);
      const T rcn = dev_rsqrt(f.c2);
      const T ang = dev_acos(f.d * dev_rsqrt(f.l1 * f.l2));
      const T s = ang * rcn * rcn;
      // entry k: cot at j_next, angle * grad phi_p (p = j_k) = e_next x c / |c|^2
      acc[k].x += (f.l2 - f.d) * rcn;
      acc[k].y += s * (en[1] * f.cz - en[2] * f.cy); acc[k].z += s * (en[2] * f.cx - en[0] * f.cz); acc[k].w += s * (en[0] * f.cy - en[1] * f.cx);
      // entry next: cot at j_k, angle * grad phi_q (q = j_next) = c x e_k / |c|^2
      T4 nx;
      nx.x = (f.l1 - f.d) * rcn;
      nx.y = s * (f.cy * e[k][2] - f.cz * e[k][1]); nx.z = s * (f.cz * e[k][0] - f.cx * e[k][2]); nx.w = s * (f.cx * e[k][1] - f.cy * e[k][0]);
      if (wrap) { acc[0].x += nx.x; acc[0].y += nx.y; acc[0].z += nx.z; acc[0].w += nx.w; }
      else { acc[kn].x += nx.x; acc[kn].y += nx.y; acc[kn].z += nx.z; acc[kn].w += nx.w; }
      theta += ang; area3 += f.c2 * rcn * (T)(1.0 / 6.0);
    }
    if (real) maxe2 = fmax(maxe2, e[k][0] * e[k][0] + e[k][1] * e[k][1] + e[k][2] * e[k][2]);
  }
  const RowScale<T> rs = row_scale<T>(area3, theta);
#pragma unroll
  for (int k = 0; k < W; k++)
    if ((uint)k < width) coef[at(k)] = entry_scale<T>(acc[k], rs);
  return dev_sqrt(maxe2);
}

// assemble_row_fan + one time step of the row, fused (BASELINE config 5: re-assembly every step): the row's
// coefficients (w, Cx, Cy, Cz) are applied to the neighbours' state straight from the registers they were
// assembled in -- lap f = sum w (f_j - f_v), G f = sum C (f_j - f_v), then reaction_update, the arithmetic
// of sell_step_kernel entry by entry in the same order -- so the 96-B row of the operator does not make the
// round trip through HBM between an assembly kernel and a step kernel (STORE = true writes it as well, for a
// caller that keeps stepping on it).  idx[] = the row ids of the neighbours (the sliced-ELL column array).
// IDENT: the plan's rows are the mesh's vertices in order (a whole-mesh plan), so the neighbour's mesh id in the table
// IS its row id and the column array is not read at all (24 B per row less).
template <typename T, int W, bool STORE, bool IDENT>
__device__ __forceinline__ T assemble_row_fan_step(const typename Vec<T>::t4 *__restrict__ pos, const uint2 *__restrict__ topo,
                                                   const uint *__restrict__ idx, typename Vec<T>::t4 *__restrict__ coef,
                                                   const typename Vec<T>::t2 *__restrict__ src, uint vg, uint row, size_t first,
                                                   uint width, const ModelT<T> &m, typename Vec<T>::t2 &out) {
  typedef typename Vec<T>::t4 T4;
  typedef typename Vec<T>::t2 T2;
  auto at = [&](int k) { return first + (size_t)k * 64; };
  const T4 P = pos[vg];
  const T2 self = src[row];
  T area3 = 0, theta = 0, maxe2 = 0;
  uint2 t[W];                                        // loads unconditional, as in assemble_row
  uint col[W];
#pragma unroll
  for (int k = 0; k < W; k++) {
    const size_t e = at((uint)k < width ? k : 0);
    const uint2 q = topo[e];
    t[k].x = q.x;
    t[k].y = (uint)k < width ? q.y : 0xffffu;
    col[k] = IDENT ? q.x : idx[e];
  }
  T e[W][3];
  T4 acc[W];
  T2 f[W];
#pragma unroll
  for (int k = 0; k < W; k++) {
    const T4 J = pos[t[k].x];
    f[k] = src[col[k]];
    e[k][0] = J.x - P.x; e[k][1] = J.y - P.y; e[k][2] = J.z - P.z;
    acc[k].x = acc[k].y = acc[k].z = acc[k].w = 0;
  }
#pragma unroll
  for (int k = 0; k < W; k++) {
    const uint s1 = t[k].y & 0xffu;
    const bool real = (t[k].y >> 16) & 1u, has1 = real && s1 < (uint)W;
    const bool wrap = (k + 1 == W) || s1 == 0u;
    const int kn = k + 1 < W ? k + 1 : 0;
    T en[3];
#pragma unroll
    for (int d = 0; d < 3; d++) en[d] = wrap ? e[0][d] : e[kn][d];
    if (has1) {
      const FanFace<T> ff = fan_face<T>(e[k], en);
      const T rcn = dev_rsqrt(ff.c2);
      const T ang = dev_acos(ff.d * dev_rsqrt(ff.l1 * ff.l2));
      const T s = ang * rcn * rcn;
      acc[k].x += (ff.l2 - ff.d) * rcn;
      acc[k].y += s * (en[1] * ff.cz - en[2] * ff.cy); acc[k].z += s * (en[2] * ff.cx - en[0] * ff.cz); acc[k].w += s * (en[0] * ff.cy - en[1] * ff.cx);
      T4 nx;
      nx.x = (ff.l1 - ff.d) * rcn;
      nx.y = s * (ff.cy * e[k][2] - ff.cz * e[k][1]); nx.z = s * (ff.cz * e[k][0] - ff.cx * e[k][2]); nx.w = s * (ff.cx * e[k][1] - ff.cy * e[k][0]);
      if (wrap) { acc[0].x += nx.x; acc[0].y += nx.y; acc[0].z += nx.z; acc[0].w += nx.w; }
      else { acc[kn].x += nx.x; acc[kn].y += nx.y; acc[kn].z += nx.z; acc[kn].w += nx.w; }
      theta += ang; area3 += ff.c2 * rcn * (T)(1.0 / 6.0);
    }
    if (real) maxe2 = fmax(maxe2, e[k][0] * e[k][0] + e[k][1] * e[k][1] + e[k][2] * e[k][2]);
  }
  const RowScale<T> rs = row_scale<T>(area3, theta);
  T lapn = 0, lapc = 0, gnx = 0, gny = 0, gnz = 0, gcx = 0, gcy = 0, gcz = 0;
#pragma unroll
  for (int k = 0; k < W; k++) {
    if ((uint)k < width) {
      const T4 c = entry_scale<T>(acc[k], rs);
      if (STORE) coef[at(k)] = c;
      const T dn = f[k].x - self.x, dc = f[k].y - self.y;
      lapn += c.x * dn; lapc += c.x * dc;
      gnx += c.y * dn; gny += c.z * dn; gnz += c.w * dn;
      gcx += c.y * dc; gcy += c.z * dc; gcz += c.w * dc;
    }
  }
  out = reaction_update<T>(self, lapn, lapc, gnx, gny, gnz, gcx, gcy, gcz, m);
  return dev_sqrt(maxe2);
}

template <typename T>
__device__ __forceinline__ T assemble_row_wide(const typename Vec<T>::t4 *__restrict__ pos, const uint2 *__restrict__ topo,
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

// PART: 0 = every row; 1 = rows of at most 6 entries only (small register footprint: 5 waves per SIMD
// in flight under the dependent loads slice -> table -> positions); 2 = the wider rows only
template <typename T, bool FAN, int PART = 0>
__global__ __launch_bounds__(256) void assemble_rows_sell_kernel(const typename Vec<T>::t4 *__restrict__ pos,
                                                                 const uint2 *__restrict__ topo, const uint *__restrict__ slice_off,
                                                                 const uint *__restrict__ rowg,
                                                                 typename Vec<T>::t4 *__restrict__ coef, uint nslices, uint V,
                                                                 double *__restrict__ blockMax) {
  const uint slice = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  T me = 0;
  if (slice < nslices) {
    const uint off = slice_off[slice], width = (slice_off[slice + 1] - off) >> 6;
    const uint row = min(slice * 64 + lane, V - 1);    // rows >= V: padding entries only (flag 0)
    const uint vg = rowg[row];
    if (width == 0) {
      // a slice of isolated vertices: no entries
    } else if (width <= 6) {
      if constexpr (PART != 2)
        me = FAN ? assemble_row_fan<T, 6>(pos, topo, coef, vg, (size_t)off + lane, 64, width)
                 : assemble_row<T, 6>(pos, topo, coef, vg, (size_t)off + lane, 64, width);
    } else if constexpr (PART != 1) {
      if (width <= 12) me = FAN ? assemble_row_fan<T, 12>(pos, topo, coef, vg, (size_t)off + lane, 64, width)
                                : assemble_row<T, 12>(pos, topo, coef, vg, (size_t)off + lane, 64, width);
      else me = assemble_row<T, 0, (1 << 30)>(pos, topo, coef, vg, (size_t)off + lane, 64, width);   // no split: all entries strided
    }
  }
  block_max_256((double)me, blockMax);
}

template <typename T, bool FAN, int PART = 0>
__global__ __launch_bounds__(256) void assemble_rows_tile_kernel(const typename Vec<T>::t4 *__restrict__ pos,
                                                                 const uint2 *__restrict__ topo, const uint *__restrict__ meta,
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
    T e = 0;
    if (nx == 0) {
      if constexpr (PART != 2)
        e = FAN ? assemble_row_fan<T, kTileWDev>(pos, topo, coef, vg, first, rpad, W)
                : assemble_row<T, kTileWDev>(pos, topo, coef, vg, first, rpad, W);
    } else if constexpr (PART != 1) {
      if (nx <= 6) e = FAN ? assemble_row_fan<T, 12, kTileWDev>(pos, topo, coef, vg, first, rpad, W + nx, xfirst)
                           : assemble_row<T, 12, kTileWDev>(pos, topo, coef, vg, first, rpad, W + nx, xfirst);
      else e = assemble_row_wide<T>(pos, topo, coef, vg, first, rpad, W + nx, xfirst);
    }
    me = fmax(me, e);
  }
  block_max_256((double)me, blockMax);
}

// Fused assembly + step over a sliced-ELL plan whose rows are all fans of at most 12 entries (regular and mildly
// irregular meshes); one wave64 per slice like sell_step_kernel / assemble_rows_sell_kernel.
// MAXW = 6: plans without a row of more than 6 entries (small register footprint, as PART = 1 above); 12: any
template <typename T, bool STORE, int MAXW, bool IDENT>
__global__ __launch_bounds__(256) void assemble_step_sell_kernel(const typename Vec<T>::t4 *__restrict__ pos,
                                                                 const uint2 *__restrict__ topo, const uint *__restrict__ slice_off,
                                                                 const uint *__restrict__ idx, const uint *__restrict__ rowg,
                                                                 typename Vec<T>::t4 *__restrict__ coef,
                                                                 const typename Vec<T>::t2 *__restrict__ src,
                                                                 typename Vec<T>::t2 *__restrict__ dst, uint nslices, uint V,
                                                                 ModelT<T> m, double *__restrict__ blockMax) {
  typedef typename Vec<T>::t2 T2;
  const uint slice = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  T me = 0;
  if (slice < nslices) {
    const uint off = slice_off[slice], width = (slice_off[slice + 1] - off) >> 6;
    const uint v = slice * 64 + lane, row = min(v, V - 1);
    const uint vg = rowg[row];
    T2 o = src[row];
    if (width == 0) {
      o = reaction_update<T>(o, 0, 0, 0, 0, 0, 0, 0, 0, m);            // a slice of isolated vertices: reaction only
    } else if (MAXW <= 6 || width <= 6) {
      me = assemble_row_fan_step<T, 6, STORE, IDENT>(pos, topo, idx, coef, src, vg, row, (size_t)off + lane, width, m, o);
    } else if constexpr (MAXW > 6) {
      me = assemble_row_fan_step<T, 12, STORE, IDENT>(pos, topo, idx, coef, src, vg, row, (size_t)off + lane, width, m, o);
    }
    if (v < V) dst[v] = o;
  }
  block_max_256((double)me, blockMax);
}

// the same displacement for the vertices of a plan's rows only (a shard's plan: rowg[r] = mesh vertex of local row r;
// every position its rows' faces read belongs to one of its rows, so the other ranks' vertices stay untouched)
template <typename T>
__global__ __launch_bounds__(256) void deform_rows_kernel(const typename Vec<T>::t4 *__restrict__ base,
                                                          typename Vec<T>::t4 *__restrict__ pos, const uint *__restrict__ rowg,
                                                          uint nrows, T amp, T q, T phase) {
  const uint r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const uint i = rowg[r];
  typename Vec<T>::t4 p = base[i];
  const T s = (T)1 + amp * sin(q * p.x + phase) * cos(q * p.y - (T)0.5 * phase) * sin(q * p.z + (T)0.25 * phase);
  p.x *= s; p.y *= s; p.z *= s;
  pos[i] = p;
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


