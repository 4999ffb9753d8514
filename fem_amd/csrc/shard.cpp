// shard.cpp -- splits an ordered (partitioned + renumbered) mesh over the ranks of one node.
// No counterpart in the reference (single GPU): rank r owns the contiguous
// partition range [r n / R, (r+1) n / R), i.e. the vertex range node_parts[...] that
// partitioning_common.cpp:99-100,132-136 makes contiguous, plus ghost copies of every vertex
// within `krings` edges of it.  With k rings a rank can take k time steps between exchanges
// (ghost rows are recomputed redundantly; ring j is valid for k - j steps).
#include <algorithm>

#include "fem_host.hpp"

namespace fem {

namespace {
// ghosts of the vertex range [b, e): all vertices outside it within k edges, ascending
void ghostsOf(const SimpleTriMesh &m, uint b, uint e, int k, std::vector<uint8_t> &mark, std::vector<uint> &out) {
  out.clear();
  std::vector<uint> frontier, next;
  for (uint v = b; v < e; v++)
    for (int i = m.rowptr[v]; i < m.rowptr[v + 1]; i++) {
      const uint u = m.col[i];
      if ((u < b || u >= e) && !mark[u]) { mark[u] = 1; frontier.push_back(u); }
    }
  out = frontier;
  for (int ring = 2; ring <= k && !frontier.empty(); ring++) {
    next.clear();
    for (uint v : frontier)
      for (int i = m.rowptr[v]; i < m.rowptr[v + 1]; i++) {
        const uint u = m.col[i];
        if ((u < b || u >= e) && !mark[u]) { mark[u] = 1; next.push_back(u); }
      }
    out.insert(out.end(), next.begin(), next.end());
    frontier.swap(next);
  }
  for (uint u : out) mark[u] = 0;
  std::sort(out.begin(), out.end());
}
}  // namespace

int buildShard(const SimpleTriMesh &m, int n, const uint *node_parts, int rank, int nranks, int krings, Shard &o) {
  const uint V = m.n_vertices();
  if (m.H == 0 || n < nranks || nranks < 1 || rank < 0 || rank >= nranks || krings < 1 || node_parts[n] != V)
    return FEM_ERR_INVALID;
  auto rangeOf = [&](int r, uint &b, uint &e) {
    b = node_parts[(int64_t)r * n / nranks];
    e = node_parts[(int64_t)(r + 1) * n / nranks];
  };
  o = Shard();
  o.rank = rank; o.nranks = nranks; o.krings = krings;
  rangeOf(rank, o.own_begin, o.own_end);
  o.send_counts.assign(nranks, 0); o.recv_counts.assign(nranks, 0);
  std::vector<uint8_t> mark(V, 0);
  std::vector<uint> ghosts;
  for (int r = 0; r < nranks; r++) {
    uint b, e;
    rangeOf(r, b, e);
    ghostsOf(m, b, e, krings, mark, ghosts);
    if (r == rank) {
      o.loc2glob.resize(o.own_end - o.own_begin);
      for (uint v = o.own_begin; v < o.own_end; v++) o.loc2glob[v - o.own_begin] = v;
      o.loc2glob.insert(o.loc2glob.end(), ghosts.begin(), ghosts.end());
      for (int q = 0; q < nranks; q++) {          // ghosts are ascending, owners' ranges are contiguous
        uint qb, qe;
        rangeOf(q, qb, qe);
        o.recv_counts[q] = (int)(std::lower_bound(ghosts.begin(), ghosts.end(), qe) -
                                 std::lower_bound(ghosts.begin(), ghosts.end(), qb));
      }
    } else {                                       // what rank r needs from me, ascending global id
      auto lo = std::lower_bound(ghosts.begin(), ghosts.end(), o.own_begin);
      auto hi = std::lower_bound(ghosts.begin(), ghosts.end(), o.own_end);
      o.send_counts[r] = (int)(hi - lo);
      for (auto it = lo; it != hi; ++it) o.send_idx.push_back(*it - o.own_begin);
    }
  }
  return FEM_OK;
}

}  // namespace fem
