// pack.cpp -- device-layout packing.
//   packRefLayout : the reference's padded SoA arrays, initializeGPUData,
//                   reference src/main.cpp:207-342 (index formulas: SURVEY.md App. A)
//   packSellLayout: the native layout -- sliced ELL, 64 rows per slice, the
//                   vertex-gradient operator folded into per-nonzero 3-vectors
#include <algorithm>
#include <cstring>

#include "fem_host.hpp"

namespace fem {

int packRefLayout(const SimpleTriMesh &m, const MeshStats &s, int n, const uint *node_parts,
                  const uint *element_parts, RefLayout &o) {
  const uint V = m.n_vertices(), F = m.n_faces();
  if (m.H == 0 || s.precision < 0 || n < 1) return FEM_ERR_INVALID;
  if (node_parts[0] != 0 || node_parts[n] != V || element_parts[n + 1] != F) return FEM_ERR_INVALID;
  o = RefLayout();
  o.n = n;

  // halo faces (main.cpp:209-232): faces touching a partition's vertices but
  // outside its own face range, ascending and unique per partition
  std::vector<int32_t> vfPtr(V + 2, 0);
  for (size_t i = 0; i < 3 * (size_t)F; i++) vfPtr[m.faces[i] + 2]++;
  for (uint i = 0; i < V; i++) vfPtr[i + 2] += vfPtr[i + 1];
  std::vector<uint> vf(3 * (size_t)F);
  for (size_t i = 0; i < 3 * (size_t)F; i++) vf[vfPtr[m.faces[i] + 1]++] = (uint)(i / 3);
  o.halo_faces_keys.assign(n + 1, 0);
  for (int p = 0; p < n; p++) {
    if (node_parts[p + 1] < node_parts[p] || element_parts[p + 1] < element_parts[p]) return FEM_ERR_INVALID;
    const size_t start = o.halo_faces.size();
    for (uint j = node_parts[p]; j < node_parts[p + 1]; j++)
      for (int e = vfPtr[j]; e < vfPtr[j + 1]; e++) {
        const uint f = vf[e];
        if (f < element_parts[p] || f >= element_parts[p + 1]) o.halo_faces.push_back(f);
      }
    std::sort(o.halo_faces.begin() + start, o.halo_faces.end());
    o.halo_faces.erase(std::unique(o.halo_faces.begin() + start, o.halo_faces.end()), o.halo_faces.end());
    o.halo_faces_keys[p + 1] = (uint)o.halo_faces.size();
  }

  // sizes (main.cpp:241-250,261-270,290-294)
  o.block_face_count.resize(n);
  for (int p = 0; p < n; p++) {
    o.block_face_count[p] = element_parts[p + 1] - element_parts[p] + o.halo_faces_keys[p + 1] - o.halo_faces_keys[p];
    o.max_vertex_count = std::max(o.max_vertex_count, node_parts[p + 1] - node_parts[p]);
    o.max_face_count = std::max(o.max_face_count, o.block_face_count[p]);
  }
  for (uint v = 0; v < V; v++) o.vv_max_neighbors = std::max(o.vv_max_neighbors, (uint)(m.rowptr[v + 1] - m.rowptr[v]));
  const size_t maxV = o.max_vertex_count, W = o.vv_max_neighbors, maxF = o.max_face_count;

  o.vertex_weights.assign((size_t)n * maxV, 0.0);
  o.nbr_v.assign((size_t)n * (W + 1) * maxV, 0u);
  o.vv_weights.assign((size_t)n * W * maxV, 0.0);
  o.face_vertices.assign((size_t)n * 3 * maxF, 0u);
  o.fv_weights_new.assign((size_t)n * 3 * maxF, 0.0);
  fem_float4 zero4 = {0, 0, 0, 0};
  o.he_grads.assign((size_t)n * 2 * maxF, zero4);

  for (int p = 0; p < n; p++) {
    // vertices: diagonal weight + ELL row (main.cpp:252-287)
    for (uint j = 0; j < node_parts[p + 1] - node_parts[p]; j++) {
      const uint v = node_parts[p] + j;
      o.vertex_weights[(size_t)p * maxV + j] = s.wVtx[v];
      uint counter = 0;
      for (int e = m.rowptr[v]; e < m.rowptr[v + 1]; e++, counter++) {
        o.nbr_v[(size_t)p * (W + 1) * maxV + (size_t)(counter + 1) * maxV + j] = m.col[e];
        o.vv_weights[(size_t)p * W * maxV + (size_t)counter * maxV + j] = s.wHej[m.row_he[e]];
      }
      o.nbr_v[(size_t)p * (W + 1) * maxV + j] = counter;
    }
    // faces: interior first, then halo (main.cpp:299-338)
    const uint nint = element_parts[p + 1] - element_parts[p];
    for (uint slot = 0; slot < o.block_face_count[p]; slot++) {
      const uint f = slot < nint ? element_parts[p] + slot : o.halo_faces[o.halo_faces_keys[p] + (slot - nint)];
      fem_float4 &g12 = o.he_grads[(size_t)p * 2 * maxF + slot];
      fem_float4 &g13 = o.he_grads[(size_t)p * 2 * maxF + maxF + slot];
      g12.x = (float)s.gradVec12[3 * f]; g12.y = (float)s.gradVec12[3 * f + 1]; g12.z = (float)s.gradVec12[3 * f + 2];
      g13.x = (float)s.gradVec13[3 * f]; g13.y = (float)s.gradVec13[3 * f + 1]; g13.z = (float)s.gradVec13[3 * f + 2];
      for (int c = 0; c < 3; c++) {   // fh iteration yields to-vertices a, b, c; its half-edge ends there
        o.face_vertices[(size_t)p * 3 * maxF + (size_t)c * maxF + slot] = m.faces[3 * f + c];
        o.fv_weights_new[(size_t)p * 3 * maxF + (size_t)c * maxF + slot] = s.meshAngle[3 * f + (c + 2) % 3];
      }
    }
  }
  return FEM_OK;
}

// Folded operator rows.  Row v, entry for neighbour j (half-edge h: v -> j):
//   w     = wHej[h]                                  (Laplacian, difference form)
//   C_vj  = (1/Theta_v) * sum over the <=2 faces f on edge (v,j) of
//           theta_{v,f} * grad(phi_j) on f           (vertex gradient, folded)
// so that  lap f (v) = sum_j w (f_j - f_v)  and  G f (v) = sum_j C_vj (f_j - f_v),
// which equal the reference's fn*wVtx + sum fn_j*wHej (wVtx = -sum wHej,
// mesh_function.cpp:263-265) and its angle-weighted mean of face gradients
// (mesh_function.h:35-110) because the hat gradients of a face sum to zero.
// With a row list (loc2glob[nloc], a shard's owned vertices followed by its ghosts) row r holds
// the operator row of global vertex loc2glob[r] with column ids remapped to local ones;
// neighbours outside the list are dropped (only the outermost ghost ring has any, and its rows
// are never consumed: DESIGN.md "Multi-GPU").
// CSR entries [b, e) of vertex v's outgoing half-edges in counter-clockwise fan order: from the
// edge with no face on its right (open fan) or the smallest neighbour id (closed fan), each next
// edge goes to the vertex opposite in the face on the left.  Entries a second fan at the same
// vertex would add (non-manifold vertex) follow in CSR order; the row then fails the fan check.
static void fanOrder(const SimpleTriMesh &m, uint v, int H3, std::vector<int> &order, std::vector<char> &used) {
  const int b = m.rowptr[v], e = m.rowptr[v + 1], n = e - b;
  order.clear();
  used.assign((size_t)n, 0);
  int q = b;
  for (int c = b; c < e; c++)
    if (m.opp[m.row_he[c]] >= H3) { q = c; break; }
  for (int cnt = 0; cnt < n; cnt++) {
    order.push_back(q);
    used[(size_t)(q - b)] = 1;
    const int h = m.row_he[q];
    if (h >= H3) break;                                       // no face on the left: the open fan ends here
    const uint o1 = m.faces[SimpleTriMesh::prev_halfedge(h)];
    int nq = -1;
    for (int c = b; c < e; c++)
      if (m.col[c] == o1) { nq = c; break; }
    if (nq < 0 || used[(size_t)(nq - b)]) break;
    q = nq;
  }
  for (int c = b; c < e; c++)
    if (!used[(size_t)(c - b)]) order.push_back(c);
}

int buildRowOp(const SimpleTriMesh &m, const MeshStats &s, RowOp &o, const uint *loc2glob, uint nloc, bool fan) {
  const uint Vg = m.n_vertices();
  const int H3 = 3 * (int)m.n_faces();
  if (m.H == 0 || s.precision < 0 || Vg == 0) return FEM_ERR_INVALID;
  const uint V = loc2glob ? nloc : Vg;
  if (V == 0) return FEM_ERR_INVALID;
  std::vector<int32_t> glob2loc;
  if (loc2glob) {
    glob2loc.assign(Vg, -1);
    for (uint r = 0; r < nloc; r++) {
      if (loc2glob[r] >= Vg || glob2loc[loc2glob[r]] >= 0) return FEM_ERR_INVALID;
      glob2loc[loc2glob[r]] = (int32_t)r;
    }
  }
  o = RowOp();
  o.V = V;
  o.rowptr.assign((size_t)V + 1, 0);
  size_t nnzMax = 0;
  for (uint r = 0; r < V; r++) { const uint v = loc2glob ? loc2glob[r] : r; nnzMax += m.rowptr[v + 1] - m.rowptr[v]; }
  if (nnzMax > 0xffffffffull) return FEM_ERR_LIMIT;
  o.col.reserve(nnzMax);
  o.nb.reserve(2 * nnzMax);
  std::vector<uint> opp2;                          // per entry: mesh id of the vertex opposite in the face left of j->v
  opp2.reserve(nnzMax);
  o.rowg.resize(V);
  o.coef.reserve(4 * nnzMax);

  auto hatGrad = [&](int f, int corner, double *g) {   // corner 0 (a): gradVec12, 1 (b): gradVec13, 2 (c): -(both)
    for (int k = 0; k < 3; k++) {
      const double g12 = s.gradVec12[3 * (size_t)f + k], g13 = s.gradVec13[3 * (size_t)f + k];
      g[k] = corner == 0 ? g12 : (corner == 1 ? g13 : -(g12 + g13));
    }
  };

  std::vector<int> order;
  std::vector<char> used;
  bool allFans = fan;
  for (uint row = 0; row < V; row++) {
    const uint v = loc2glob ? loc2glob[row] : row;
    o.rowg[row] = v;
    double theta = 0;
    for (int e = m.rowptr[v]; e < m.rowptr[v + 1]; e++) {
      const int h = m.row_he[e];
      if (h < H3) theta += s.meshAngle[SimpleTriMesh::prev_halfedge(h)];
    }
    const size_t rowStart = o.col.size();
    if (fan) fanOrder(m, v, H3, order, used);
    for (int ei = 0, ne = m.rowptr[v + 1] - m.rowptr[v]; ei < ne; ei++) {
      const int e = fan ? order[(size_t)ei] : m.rowptr[v] + ei;
      uint lc = m.col[e];
      if (loc2glob) {
        const int32_t l = glob2loc[m.col[e]];
        if (l < 0) continue;                       // neighbour outside the shard
        lc = (uint)l;
      }
      const int h = m.row_he[e], ho = m.opp[h];
      double C[3] = {0, 0, 0}, g[3];
      if (theta > 0) {
        if (h < H3) {           // face left of v->j: v = corner h%3, j = next corner
          hatGrad(h / 3, (h % 3 + 1) % 3, g);
          const double a = s.meshAngle[SimpleTriMesh::prev_halfedge(h)];
          for (int d = 0; d < 3; d++) C[d] += a * g[d];
        }
        if (ho < H3) {          // face left of j->v: j = corner ho%3, angle at v = meshAngle[ho]
          hatGrad(ho / 3, ho % 3, g);
          const double a = s.meshAngle[ho];
          for (int d = 0; d < 3; d++) C[d] += a * g[d];
        }
        for (int d = 0; d < 3; d++) C[d] /= theta;
      }
      o.col.push_back(lc);
      // device assembly: the neighbour's mesh id now; the vertices opposite the edge (v, j) in the
      // faces left of v->j and of j->v are neighbours of v too -- parked as mesh ids, turned into
      // row slots below
      o.nb.push_back(m.col[e]);
      o.nb.push_back(h < H3 ? m.faces[SimpleTriMesh::prev_halfedge(h)] : 0xffffffffu);
      opp2.push_back(ho < H3 ? m.faces[SimpleTriMesh::prev_halfedge(ho)] : 0xffffffffu);
      o.coef.push_back(s.wHej[h]);
      o.coef.push_back(C[0]); o.coef.push_back(C[1]); o.coef.push_back(C[2]);
    }
    // (j, o1, o2) -> (j, slot(o1) | slot(o2) << 8 | 1 << 16): slot = position of that neighbour among
    // this row's entries, 0xff = no face on that side (or the neighbour lies outside the shard)
    const size_t rowEnd = o.col.size();
    if (rowEnd - rowStart > 254) return FEM_ERR_LIMIT;
    for (size_t e = rowStart; e < rowEnd; e++) {
      uint slot[2] = {0xffu, 0xffu};
      const uint want[2] = {o.nb[2 * e + 1], opp2[e]};
      for (int sd = 0; sd < 2; sd++)
        if (want[sd] != 0xffffffffu)
          for (size_t q = rowStart; q < rowEnd; q++)
            if (o.nb[2 * q] == want[sd]) { slot[sd] = (uint)(q - rowStart); break; }
      o.nb[2 * e + 1] = slot[0] | (slot[1] << 8) | (1u << 16);
    }
    if (allFans) {
      // one fan in order: a left face leads to the next entry (or, from the last one, back to entry 0),
      // and that entry's right face leads back here; as many right faces as left faces
      const uint n = (uint)(rowEnd - rowStart);
      uint left = 0, right = 0;
      for (uint k = 0; k < n && allFans; k++) {
        const uint w = o.nb[2 * (rowStart + k) + 1], s1 = w & 0xffu, s2 = (w >> 8) & 0xffu;
        if (s2 != 0xffu) right++;
        if (s1 == 0xffu) continue;
        left++;
        const bool next = s1 == k + 1 && s1 < n, wrap = s1 == 0 && k == n - 1 && k > 0;
        allFans = (next || wrap) && ((o.nb[2 * (rowStart + s1) + 1] >> 8) & 0xffu) == k;
      }
      allFans = allFans && left == right;
    }
    o.rowptr[row + 1] = (uint)o.col.size();
  }
  o.fanOrdered = allFans;
  return FEM_OK;
}

// Native layout 1: sliced ELL, 64 rows per slice; rows padded to the slice width with
// (idx = own row, coef = 0).
int packSellLayout(const RowOp &op, SellLayout &o) {
  const uint V = op.V;
  o = SellLayout();
  o.V = V;
  o.nslices = (V + 63) / 64;
  o.slice_off.assign(o.nslices + 1, 0);
  for (uint sl = 0; sl < o.nslices; sl++) {
    uint w = 0;
    for (uint v = sl * 64; v < std::min(V, sl * 64 + 64); v++) w = std::max(w, op.rowptr[v + 1] - op.rowptr[v]);
    uint64_t next = (uint64_t)o.slice_off[sl] + (uint64_t)w * 64;
    if (next > 0xffffffffull) return FEM_ERR_LIMIT;
    o.slice_off[sl + 1] = (uint)next;
  }
  const size_t entries = o.slice_off[o.nslices];
  o.idx.resize(entries);
  o.coef.assign(4 * entries, 0.0);
  o.topo.assign(2 * entries, 0u);
  for (uint sl = 0; sl < o.nslices; sl++) {
    const uint off = o.slice_off[sl], width = (o.slice_off[sl + 1] - off) / 64;
    for (uint lane = 0; lane < 64; lane++) {
      const uint row = sl * 64 + lane;
      const uint self = row < V ? row : V - 1;
      for (uint k = 0; k < width; k++) o.idx[off + (size_t)k * 64 + lane] = self;
      if (row >= V) continue;
      uint k = 0;
      for (uint e = op.rowptr[row]; e < op.rowptr[row + 1]; e++, k++) {
        const size_t pos = off + (size_t)k * 64 + lane;
        o.idx[pos] = op.col[e];
        o.topo[2 * pos] = op.nb[2 * (size_t)e]; o.topo[2 * pos + 1] = op.nb[2 * (size_t)e + 1];
        for (int d = 0; d < 4; d++) o.coef[4 * pos + d] = op.coef[4 * (size_t)e + d];
      }
    }
  }
  return FEM_OK;
}

// Native layout 2: tiles for the fused k-step kernel.  A tile = a contiguous range of rows (a
// partition, or a piece of one when the partition does not fit) that the kernel advances k steps
// in one launch, plus the k rings of rows around it: local list = owned rows, then ring 1, ...,
// ring k (each ring: rows wider than W first, widest first, then ascending).  m[j] = number of local rows within j edges of the owned range.
// The kernel keeps the state of all m[k] local rows in LDS and the operator rows of the first
// m[k-1] in registers (thread t holds rows t, t+T, ..., at most R of them), so a tile must
// satisfy m[k-1] <= T*R and m[k] <= maxLocal; ranges that do not are split in halves.
// Per tile: meta[16] = {base, ghost_off, row_off, rows_pad, k, m[0..k]};
//   ghost[ghost_off + i]            = row id of local row m[0]+i               (i < m[k]-m[0])
//   idx2[(row_off + (w/2)*rows_pad*2)/2 ...]  see below
//   coef[row_off + w*rows_pad + r]  = (w, Cx, Cy, Cz) of entry w of local row r (r < m[k-1])
//   idx2[row_off/2 + (w/2)*rows_pad + r] = local ids of entries w, w+1 of row r, 16 bits each
// (padding entries: own local id, coef 0).  rows_pad = m[k-1] rounded up to 64; row_off is even.
// Rows with more than W entries (irregular meshes) keep their first W in the register tile; the
// rest ("extras", in row order) follow the tile's block: coef/topo[row_off + W*rows_pad + e], and
//   xinfo[meta[14] + r]            = first extra of row r | count << 16          (r < rows_pad)
//   xinfo[meta[14] + rows_pad + e] = local column id of extra e                  (e < meta[15])
// meta[14], meta[15] = xinfo offset, number of extras; tiles without extras have no xinfo block.
// A tile holds at most maxExtra extras and a row at most 65535 of them.
int packTileLayout(const RowOp &op, const std::vector<std::pair<uint, uint>> &ranges, int ksteps, uint T, uint R,
                   uint W, uint maxLocal, uint maxExtra, TileLayout &o) {
  if (ksteps < 1 || ksteps > 8 || (W & 1) || W == 0 || T == 0 || R == 0 || maxLocal > 65535 || maxExtra > 65535)
    return FEM_ERR_INVALID;
  const uint V = op.V;
  o = TileLayout();
  o.ksteps = ksteps; o.T = T; o.R = R; o.W = W;
  std::vector<std::pair<uint, uint>> work(ranges.rbegin(), ranges.rend());
  std::vector<uint> stamp(V, 0xffffffffu), local;
  std::vector<uint> locId(V, 0);
  uint tileNo = 0;
  while (!work.empty()) {
    const uint b = work.back().first, e = work.back().second;
    work.pop_back();
    if (e <= b) continue;
    // rings by breadth-first search from the owned range
    local.clear();
    uint m[9] = {0};
    for (uint v = b; v < e; v++) { stamp[v] = tileNo; local.push_back(v); }
    m[0] = e - b;
    size_t frontB = 0;
    bool fits = true;
    for (int ring = 1; ring <= ksteps; ring++) {
      const size_t frontE = local.size();
      for (size_t i = frontB; i < frontE; i++) {
        const uint v = local[i];
        for (uint q = op.rowptr[v]; q < op.rowptr[v + 1]; q++) {
          const uint u = op.col[q];
          if (stamp[u] != tileNo) { stamp[u] = tileNo; local.push_back(u); }
        }
      }
      // each ring ascending -- behind its rows of more than W entries, widest first (irregular meshes): their extras
      // run in a loop that a wave executes if ANY of its 64 rows has one, so wide rows are kept together (drivers
      // number the owned rows the same way: fem_mesh_wide_rows_first)
      std::sort(local.begin() + frontE, local.end(), [&](uint a, uint c) {
        const uint wa = std::max(op.rowptr[a + 1] - op.rowptr[a], W), wc = std::max(op.rowptr[c + 1] - op.rowptr[c], W);
        return wa != wc ? wa > wc : a < c;
      });
      frontB = frontE;
      m[ring] = (uint)local.size();
      if (ring == ksteps - 1 && m[ring] > T * R) fits = false;   // rows held in registers
      if (m[ring] > maxLocal) fits = false;                      // rows held in LDS
      if (!fits) break;
    }
    if (ksteps == 1 && m[0] > T * R) fits = false;
    uint xcount = 0;
    if (fits) {
      const uint rowsIn = m[ksteps - 1];
      for (uint r = 0; r < rowsIn; r++) {
        const uint d = op.rowptr[local[r] + 1] - op.rowptr[local[r]];
        if (d > W) xcount += d - W;
      }
      if (xcount > maxExtra) fits = false;
    }
    tileNo++;
    if (tileNo == 0xffffffffu) return FEM_ERR_LIMIT;
    if (!fits) {
      if (e - b < 2) return FEM_ERR_LIMIT;
      const uint mid = b + (e - b) / 2;
      work.push_back(std::make_pair(mid, e));
      work.push_back(std::make_pair(b, mid));
      continue;
    }
    for (uint i = 0; i < local.size(); i++) locId[local[i]] = i;
    const uint rows = m[ksteps - 1], rowsPad = (rows + 63) / 64 * 64;
    const size_t rowOff = o.coef.size() / 4;
    const size_t blockEntries = (size_t)W * rowsPad + (xcount + 1) / 2 * 2;
    if (rowOff + blockEntries > 0xffffffffull || o.xinfo.size() + rowsPad + xcount > 0xffffffffull) return FEM_ERR_LIMIT;
    uint meta[16] = {b, (uint)o.ghost.size(), (uint)rowOff, rowsPad, (uint)ksteps};
    for (int j = 0; j <= ksteps; j++) meta[5 + j] = m[j];
    const size_t xOff = o.xinfo.size();
    meta[14] = (uint)xOff; meta[15] = xcount;
    o.meta.insert(o.meta.end(), meta, meta + 16);
    o.ghost.insert(o.ghost.end(), local.begin() + m[0], local.end());
    o.coef.resize(4 * (rowOff + blockEntries), 0.0);
    o.topo.resize(2 * (rowOff + blockEntries), 0u);
    o.idx2.resize((rowOff + blockEntries) / 2, 0u);
    if (xcount) o.xinfo.resize(xOff + rowsPad + xcount, 0u);
    uint xnext = 0;
    for (uint r = 0; r < rowsPad; r++) {
      uint ids[16];
      for (uint w = 0; w < W; w++) ids[w] = r < rows ? r : 0;
      if (r < rows) {
        const uint v = local[r];
        uint w = 0;
        const uint xfirst = xnext;
        for (uint q = op.rowptr[v]; q < op.rowptr[v + 1]; q++, w++) {
          size_t pos;
          if (w < W) {
            ids[w] = locId[op.col[q]];
            pos = rowOff + (size_t)w * rowsPad + r;
          } else {
            o.xinfo[xOff + rowsPad + xnext] = locId[op.col[q]];
            pos = rowOff + (size_t)W * rowsPad + xnext++;
          }
          o.topo[2 * pos] = op.nb[2 * (size_t)q]; o.topo[2 * pos + 1] = op.nb[2 * (size_t)q + 1];
          for (int d = 0; d < 4; d++) o.coef[4 * pos + d] = op.coef[4 * (size_t)q + d];
        }
        if (xnext > xfirst) o.xinfo[xOff + r] = xfirst | ((xnext - xfirst) << 16);
      }
      for (uint w = 0; w < W; w += 2) o.idx2[rowOff / 2 + (size_t)(w / 2) * rowsPad + r] = ids[w] | (ids[w + 1] << 16);
    }
    o.maxLocal = std::max(o.maxLocal, m[ksteps]);
    o.maxExtra = std::max(o.maxExtra, xcount);
    o.ntiles++;
    o.rowSteps += rows;
  }
  o.ghost.push_back(0);            // pad: the kernel reads ghost[goff] even for tiles without rings
  o.xinfo.push_back(0);
  return FEM_OK;
}

}  // namespace fem
