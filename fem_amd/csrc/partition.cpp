// partition.cpp -- block partitions + reorder, the contract of reference
// src/partitioning_common.cpp:21-176 without METIS.
//
// Stage R0 (METIS_PartMeshDual, partitioning_common.cpp:51-58) is replaced by a
// deterministic geometric k-way split: vertices are ordered along a 3-D Hilbert
// curve through the mesh's bounding box and cut into n runs of equal length.
// Stages R1-R3 (face buckets, stable counting-sort renumbering, mesh rebuild)
// keep the reference's exact rules.
#include <algorithm>
#include <cstring>

#include "fem_host.hpp"

namespace fem {

// Hilbert index of a point on a 2^bits grid per axis (Skilling's transpose form).
static uint64_t hilbert3(uint32_t x, uint32_t y, uint32_t z, int bits) {
  uint32_t X[3] = {x, y, z};
  const uint32_t M = 1u << (bits - 1);
  for (uint32_t Q = M; Q > 1; Q >>= 1) {
    const uint32_t P = Q - 1;
    for (int i = 0; i < 3; i++) {
      if (X[i] & Q) X[0] ^= P;
      else { uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
  for (int i = 1; i < 3; i++) X[i] ^= X[i - 1];
  uint32_t t = 0;
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  for (int i = 0; i < 3; i++) X[i] ^= t;
  uint64_t key = 0;
  for (int b = bits - 1; b >= 0; b--)
    for (int i = 0; i < 3; i++) key = (key << 1) | ((X[i] >> b) & 1u);
  return key;
}

int partitionVertices(const SimpleTriMesh &m, int n, int64_t *npart) {
  const int64_t V = m.n_vertices();
  if (n < 1 || V < n) return FEM_ERR_INVALID;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = 0; i < V; i++)
    for (int k = 0; k < 3; k++) {
      lo[k] = std::min(lo[k], m.points[3 * i + k]);
      hi[k] = std::max(hi[k], m.points[3 * i + k]);
    }
  double ext = 0;
  for (int k = 0; k < 3; k++) ext = std::max(ext, hi[k] - lo[k]);
  if (!(ext > 0)) ext = 1;
  const int bits = 21;
  const double scale = (double)((1u << bits) - 1) / ext;
  std::vector<std::pair<uint64_t, uint32_t>> keys((size_t)V);
  for (int64_t i = 0; i < V; i++) {
    uint32_t q[3];
    for (int k = 0; k < 3; k++) q[k] = (uint32_t)((m.points[3 * i + k] - lo[k]) * scale);
    keys[i] = std::make_pair(hilbert3(q[0], q[1], q[2], bits), (uint32_t)i);
  }
  std::sort(keys.begin(), keys.end());              // ties broken by vertex id
  for (int64_t r = 0; r < V; r++) npart[keys[r].second] = (r * (int64_t)n) / V;
  return FEM_OK;
}

// R1, partitioning_common.cpp:78-95
void faceParts(const SimpleTriMesh &m, const int64_t *npart, int n, int64_t *epart) {
  const int64_t F = m.n_faces();
  for (int64_t i = 0; i < F; i++) {
    int64_t p1 = npart[m.faces[3 * i]], p2 = npart[m.faces[3 * i + 1]], p3 = npart[m.faces[3 * i + 2]];
    epart[i] = (p1 == p2 && p2 == p3) ? p1 : n;
  }
}

// R2, partitioning_common.cpp:112-139: part[] becomes the new position of each
// old index (stable within a part), parts[0..n] the start offsets.
int createOrderArray(uint *parts, int64_t *part, int64_t N, int n) {
  std::vector<int64_t> freq((size_t)n, 0);
  for (int64_t i = 0; i < N; i++) {
    if (part[i] < 0 || part[i] >= n) return FEM_ERR_INVALID;
    freq[part[i]]++;
  }
  std::vector<int64_t> start((size_t)n + 1, 0);
  for (int i = 0; i < n; i++) start[i + 1] = start[i] + freq[i];
  for (int i = 0; i <= n; i++) parts[i] = (uint)start[i];
  std::vector<int64_t> next(start.begin(), start.end() - 1);
  for (int64_t i = 0; i < N; i++) part[i] = next[part[i]]++;
  return FEM_OK;
}

// R3, partitioning_common.cpp:141-176: vertices and faces in the new order,
// corner order of every face preserved.
void reorderMesh(const SimpleTriMesh &original, SimpleTriMesh *ordered, const int64_t *npart,
                 const int64_t *epart) {
  const int64_t V = original.n_vertices(), F = original.n_faces();
  ordered->clean();
  ordered->points.resize(3 * (size_t)V);
  ordered->faces.resize(3 * (size_t)F);
  const bool col = original.colors.size() == original.points.size() && V > 0;
  if (col) ordered->colors.resize(3 * (size_t)V);
  for (int64_t i = 0; i < V; i++)
    for (int k = 0; k < 3; k++) {
      ordered->points[3 * npart[i] + k] = original.points[3 * i + k];
      if (col) ordered->colors[3 * npart[i] + k] = original.colors[3 * i + k];
    }
  for (int64_t i = 0; i < F; i++)
    for (int k = 0; k < 3; k++) ordered->faces[3 * epart[i] + k] = (uint)npart[original.faces[3 * i + k]];
}

// number of mesh edges whose end points lie in different parts (the value the
// reference returns is METIS's dual-graph objval; this is its vertex-graph analogue)
int64_t edgeCut(const SimpleTriMesh &m, const int64_t *npart) {
  int64_t cut = 0;
  for (int h = 0; h < m.H; h++)
    if (m.he_from[h] < m.he_to[h] && npart[m.he_from[h]] != npart[m.he_to[h]]) cut++;
  return cut;
}

int partition(const SimpleTriMesh &originalMesh, SimpleTriMesh *orderedMesh, uint **node_parts,
              uint **element_parts, int n) {
  const int64_t V = originalMesh.n_vertices(), F = originalMesh.n_faces();
  std::vector<int64_t> npart((size_t)V), epart((size_t)F);
  int rc = partitionVertices(originalMesh, n, npart.data());
  if (rc != FEM_OK) return rc;
  *node_parts = new uint[n + 1];
  *element_parts = new uint[n + 2];
  int64_t cut = originalMesh.H ? edgeCut(originalMesh, npart.data()) : 0;
  faceParts(originalMesh, npart.data(), n, epart.data());
  createOrderArray(*node_parts, npart.data(), V, n);
  createOrderArray(*element_parts, epart.data(), F, n + 1);
  reorderMesh(originalMesh, orderedMesh, npart.data(), epart.data());
  return (int)std::min<int64_t>(cut, 0x7fffffff);
}

}  // namespace fem
