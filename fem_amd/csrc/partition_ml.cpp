// partition_ml.cpp -- multilevel k-way partition of the DUAL graph of the mesh, the stage METIS_PartMeshDual
// occupies in the reference (src/partitioning_common.cpp:51-58: ncommon = 2, i.e. two elements are adjacent when
// they share an edge; only the returned node partition npart is used, :78-100).  METIS itself is not available
// (not vendored, version unpinned), so this is the published scheme it implements, written out:
//   1. coarsening by heavy-edge matching until about 24 coarse nodes per part are left (node weight = faces
//      merged, edge weight = dual edges merged, centroid carried along),
//   2. initial k-way partition of the coarsest graph by recursive bisection -- of the coarse nodes' centroids along
//      the longest axis at the weighted median (default), or coordinate-free by breadth-first graph growing from
//      pseudo-peripheral nodes, as METIS's initial partitioning is (FEM_ML_INITIAL=graph); hierarchical part
//      numbering either way: consecutive ids are neighbours on the surface, which the sharding relies on,
//   3. uncoarsening with greedy k-way boundary refinement at every level (a node moves to the adjacent part it is
//      most connected to when that lowers the cut, or keeps it and improves the balance, within the weight cap),
//   4. node partition induced from the element partition: a vertex goes to the part that holds most of its faces
//      (ties: lowest part id), then a pass that moves boundary vertices out of parts above the vertex cap.
// Deterministic: fixed visiting orders, ties by id, no random numbers, no threads.
// The caps of the reference's caller (<= 1024 vertices and faces per block, main.cpp:357-364) enter as the
// balance constraint: part weights stay within kImbalance of the mean, so choosing n >= V / (1024 / kImbalance)
// keeps every block under them.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "fem_host.hpp"

namespace fem {

namespace {

constexpr double kImbalance = 1.03;

struct Graph {
  uint32_t N = 0;
  std::vector<uint32_t> xadj, adj, ewgt, vwgt;     // CSR, edge weights, node weights
  std::vector<float> cx, cy, cz;                   // weighted centroid of the faces merged into the node
  std::vector<uint32_t> cmap;                      // fine node -> coarse node (of the next level)
};

void buildDual(const SimpleTriMesh &m, Graph &g) {
  const uint32_t F = m.n_faces();
  const int H3 = 3 * (int)F;
  g.N = F;
  g.xadj.assign((size_t)F + 1, 0); g.adj.reserve(3 * (size_t)F); g.ewgt.reserve(3 * (size_t)F);
  g.vwgt.assign(F, 1u);
  g.cx.resize(F); g.cy.resize(F); g.cz.resize(F);
  for (uint32_t f = 0; f < F; f++) {
    for (int k = 0; k < 3; k++) {
      const int ho = m.opp[3 * f + k];
      if (ho >= 0 && ho < H3 && (uint32_t)(ho / 3) != f) { g.adj.push_back((uint32_t)(ho / 3)); g.ewgt.push_back(1u); }
    }
    g.xadj[f + 1] = (uint32_t)g.adj.size();
    const double *a = &m.points[3 * (size_t)m.faces[3 * f]], *b = &m.points[3 * (size_t)m.faces[3 * f + 1]],
                 *c = &m.points[3 * (size_t)m.faces[3 * f + 2]];
    g.cx[f] = (float)((a[0] + b[0] + c[0]) / 3); g.cy[f] = (float)((a[1] + b[1] + c[1]) / 3); g.cz[f] = (float)((a[2] + b[2] + c[2]) / 3);
  }
}

// heavy-edge matching + contraction; returns false when the graph no longer shrinks usefully
bool coarsen(Graph &g, Graph &c, uint32_t maxNodeWeight) {
  const uint32_t N = g.N;
  std::vector<uint32_t> match(N, 0xffffffffu);
  g.cmap.assign(N, 0);
  // visiting order: a fixed stride co-prime with N scatters the visits over the mesh (index order would grow the
  // clusters along the numbering); the same sequence on every run
  uint64_t stride = (uint64_t)(0.6180339887 * N) | 1u;
  while (std::gcd(stride, (uint64_t)N) != 1) stride += 2;
  uint32_t nc = 0;
  uint64_t pos = 0;
  for (uint32_t i = 0; i < N; i++, pos = (pos + stride) % N) {
    const uint32_t u = (uint32_t)pos;
    if (match[u] != 0xffffffffu) continue;
    uint32_t best = 0xffffffffu, bw = 0;
    for (uint32_t e = g.xadj[u]; e < g.xadj[u + 1]; e++) {
      const uint32_t v = g.adj[e];
      if (match[v] != 0xffffffffu || g.vwgt[u] + g.vwgt[v] > maxNodeWeight) continue;
      const uint32_t w = g.ewgt[e];
      if (best == 0xffffffffu || w > bw || (w == bw && (g.vwgt[v] < g.vwgt[best] || (g.vwgt[v] == g.vwgt[best] && v < best)))) { best = v; bw = w; }
    }
    match[u] = best == 0xffffffffu ? u : best;
    if (best != 0xffffffffu) match[best] = u;
    g.cmap[u] = nc;
    if (best != 0xffffffffu) g.cmap[best] = nc;
    nc++;
  }
  if (nc > N - N / 20) return false;                 // less than 5 % fewer nodes: stop here
  c = Graph();
  c.N = nc;
  c.vwgt.assign(nc, 0); c.cx.assign(nc, 0); c.cy.assign(nc, 0); c.cz.assign(nc, 0);
  c.xadj.assign((size_t)nc + 1, 0);
  std::vector<uint32_t> first(nc, 0xffffffffu), second(nc, 0xffffffffu);
  for (uint32_t u = 0; u < N; u++) {
    const uint32_t cu = g.cmap[u];
    if (first[cu] == 0xffffffffu) first[cu] = u; else second[cu] = u;
  }
  std::vector<uint32_t> slot(nc, 0xffffffffu);       // position of coarse neighbour in the row being built
  c.adj.reserve(g.adj.size()); c.ewgt.reserve(g.adj.size());
  for (uint32_t cu = 0; cu < nc; cu++) {
    const size_t rowStart = c.adj.size();
    double wx = 0, wy = 0, wz = 0;
    for (uint32_t u : {first[cu], second[cu]}) {
      if (u == 0xffffffffu) continue;
      const double w = g.vwgt[u];
      c.vwgt[cu] += g.vwgt[u];
      wx += w * g.cx[u]; wy += w * g.cy[u]; wz += w * g.cz[u];
      for (uint32_t e = g.xadj[u]; e < g.xadj[u + 1]; e++) {
        const uint32_t cv = g.cmap[g.adj[e]];
        if (cv == cu) continue;
        if (slot[cv] != 0xffffffffu && slot[cv] >= rowStart) c.ewgt[slot[cv]] += g.ewgt[e];
        else { slot[cv] = (uint32_t)c.adj.size(); c.adj.push_back(cv); c.ewgt.push_back(g.ewgt[e]); }
      }
    }
    c.cx[cu] = (float)(wx / c.vwgt[cu]); c.cy[cu] = (float)(wy / c.vwgt[cu]); c.cz[cu] = (float)(wz / c.vwgt[cu]);
    c.xadj[cu + 1] = (uint32_t)c.adj.size();
  }
  return true;
}

// weighted recursive coordinate bisection of the coarsest graph's centroids
struct WeightedRcb {
  const Graph &g;
  std::vector<uint32_t> &part;
  std::vector<uint32_t> ids;
  void run(size_t b, size_t e, uint32_t p0, uint32_t np) {
    while (np > 1 && e - b > 1) {
      float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
      uint64_t total = 0;
      for (size_t i = b; i < e; i++) {
        const uint32_t u = ids[i];
        const float p[3] = {g.cx[u], g.cy[u], g.cz[u]};
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
        total += g.vwgt[u];
      }
      int ax = 0;
      for (int k = 1; k < 3; k++)
        if (hi[k] - lo[k] > hi[ax] - lo[ax]) ax = k;
      const std::vector<float> &C = ax == 0 ? g.cx : (ax == 1 ? g.cy : g.cz);
      std::sort(ids.begin() + b, ids.begin() + e, [&C](uint32_t x, uint32_t y) { return C[x] < C[y] || (C[x] == C[y] && x < y); });
      const uint32_t nl = np / 2;
      const uint64_t target = total * nl / np;
      uint64_t acc = 0;
      size_t mid = b;
      while (mid < e && acc + g.vwgt[ids[mid]] / 2 < target) acc += g.vwgt[ids[mid++]];
      mid = std::min(std::max(mid, b + 1), e - 1);
      run(b, mid, p0, nl);
      b = mid; p0 += nl; np -= nl;
    }
    for (size_t i = b; i < e; i++) part[ids[i]] = p0;
  }
};

// coordinate-free initial partition: recursive bisection of the coarsest graph by breadth-first growing from a
// pseudo-peripheral node (two sweeps), the first nodes up to the target weight to the left; further components of a
// disconnected subset are appended lowest id first.  Same hierarchical numbering as the coordinate bisection.
struct GraphGrow {
  const Graph &g;
  std::vector<uint32_t> &part;
  std::vector<uint32_t> ids, tag, seen, order;
  uint32_t stamp = 0;
  void bfs(uint32_t root, uint32_t t, uint32_t visit) {
    size_t head = order.size();
    order.push_back(root); seen[root] = visit;
    while (head < order.size()) {
      const uint32_t u = order[head++];
      for (uint32_t e = g.xadj[u]; e < g.xadj[u + 1]; e++) {
        const uint32_t v = g.adj[e];
        if (tag[v] == t && seen[v] != visit) { seen[v] = visit; order.push_back(v); }
      }
    }
  }
  void run(size_t b, size_t e, uint32_t p0, uint32_t np) {
    while (np > 1 && e - b > 1) {
      const uint32_t t = ++stamp;
      uint64_t total = 0;
      uint32_t root = 0xffffffffu;
      for (size_t i = b; i < e; i++) { tag[ids[i]] = t; total += g.vwgt[ids[i]]; root = std::min(root, ids[i]); }
      for (int sweep = 0; sweep < 2; sweep++) { order.clear(); bfs(root, t, ++stamp); root = order.back(); }
      order.clear();
      const uint32_t visit = ++stamp;
      bfs(root, t, visit);
      if (order.size() < e - b) {
        std::vector<uint32_t> rest(ids.begin() + b, ids.begin() + e);
        std::sort(rest.begin(), rest.end());
        for (uint32_t v : rest)
          if (seen[v] != visit) bfs(v, t, visit);
      }
      const uint32_t nl = np / 2;
      const uint64_t target = total * nl / np;
      uint64_t acc = 0;
      size_t cut = 0;
      while (cut < order.size() && acc + g.vwgt[order[cut]] / 2 < target) acc += g.vwgt[order[cut++]];
      cut = std::min(std::max<size_t>(cut, 1), order.size() - 1);
      std::copy(order.begin(), order.end(), ids.begin() + b);
      const size_t mid = b + cut;
      run(b, mid, p0, nl);
      b = mid; p0 += nl; np -= nl;
    }
    for (size_t i = b; i < e; i++) part[ids[i]] = p0;
  }
};

// greedy k-way boundary refinement (positive gain, or zero gain with better balance), weight cap maxW
void refine(const Graph &g, std::vector<uint32_t> &part, std::vector<uint64_t> &W, uint64_t maxW, uint64_t minW, int passes) {
  std::vector<uint32_t> nbParts;
  std::vector<uint32_t> nbConn;
  for (int pass = 0; pass < passes; pass++) {
    size_t moved = 0;
    for (uint32_t u = 0; u < g.N; u++) {
      const uint32_t a = part[u];
      nbParts.clear(); nbConn.clear();
      uint32_t own = 0;
      for (uint32_t e = g.xadj[u]; e < g.xadj[u + 1]; e++) {
        const uint32_t b = part[g.adj[e]];
        if (b == a) { own += g.ewgt[e]; continue; }
        size_t k = 0;
        while (k < nbParts.size() && nbParts[k] != b) k++;
        if (k == nbParts.size()) { nbParts.push_back(b); nbConn.push_back(0); }
        nbConn[k] += g.ewgt[e];
      }
      if (nbParts.empty()) continue;                    // interior node
      const uint32_t w = g.vwgt[u];
      long bestGain = -1;
      uint32_t best = 0xffffffffu;
      const bool over = W[a] > maxW;                    // an overweight part gives nodes away even at a loss
      for (size_t k = 0; k < nbParts.size(); k++) {
        const uint32_t b = nbParts[k];
        if (W[b] + w > maxW) continue;
        const long gain = (long)nbConn[k] - (long)own;
        const bool feeds = W[b] < minW && W[a] >= minW + w;      // an underweight part is fed even at a loss
        const bool ok = gain > 0 || (gain == 0 && W[b] + w < W[a]) || ((over || feeds) && W[b] + w < W[a]);
        if (!ok) continue;
        if (best == 0xffffffffu || gain > bestGain || (gain == bestGain && (W[b] < W[best] || (W[b] == W[best] && b < best)))) { best = b; bestGain = gain; }
      }
      if (best == 0xffffffffu) continue;
      if (W[a] <= w) continue;                          // never empty a part
      part[u] = best; W[a] -= w; W[best] += w;
      moved++;
    }
    if (moved == 0) break;
  }
}

}  // namespace

int64_t dualEdgeCut(const SimpleTriMesh &m, const int64_t *epart) {
  const int H3 = 3 * (int)m.n_faces();
  int64_t cut = 0;
  for (int h = 0; h < H3; h++) {
    const int ho = m.opp[h];
    if (ho > h && ho < H3 && epart[h / 3] != epart[ho / 3]) cut++;
  }
  return cut;
}

int partitionVerticesMultilevel(const SimpleTriMesh &m, int n, int64_t *npart, int64_t *epartOut) {
  const uint32_t V = m.n_vertices(), F = m.n_faces();
  if (n < 1 || m.H == 0) return FEM_ERR_INVALID;
  if (n == 1 || F == 0) { std::fill(npart, npart + V, (int64_t)0); if (epartOut) std::fill(epartOut, epartOut + F, (int64_t)0); return FEM_OK; }
  if ((uint32_t)n > F) return FEM_ERR_INVALID;

  // 1. coarsen
  std::vector<Graph> levels(1);
  buildDual(m, levels[0]);
  const uint32_t coarseTarget = std::max<uint32_t>(24u * (uint32_t)n, 64u);
  const uint32_t maxNodeWeight = std::max<uint32_t>(1u, (uint32_t)(F / (uint64_t)n / 12));   // a coarse node stays well below a part
  while (levels.back().N > coarseTarget && levels.size() < 40) {
    Graph c;
    if (!coarsen(levels.back(), c, maxNodeWeight)) break;
    levels.push_back(std::move(c));
  }
  // 2. initial partition of the coarsest level
  const Graph &top = levels.back();
  std::vector<uint32_t> part(top.N, 0);
  // Initial partition: weighted coordinate bisection of the coarse nodes' centroids by default (on the surfaces
  // measured it gives the lower cut and the rounder parts); FEM_ML_INITIAL=graph selects the
  // coordinate-free graph-growing bisection instead (surfaces whose Euclidean neighbours are geodesically far apart)
  if (!(getenv("FEM_ML_INITIAL") && !strcmp(getenv("FEM_ML_INITIAL"), "graph"))) {
    WeightedRcb rcb{top, part, std::vector<uint32_t>(top.N)};
    std::iota(rcb.ids.begin(), rcb.ids.end(), 0u);
    rcb.run(0, top.N, 0, (uint32_t)n);
  } else {
    GraphGrow gg{top, part, std::vector<uint32_t>(top.N), std::vector<uint32_t>(top.N, 0), std::vector<uint32_t>(top.N, 0), {}};
    std::iota(gg.ids.begin(), gg.ids.end(), 0u);
    gg.run(0, top.N, 0, (uint32_t)n);
  }
  const uint64_t maxW = (uint64_t)std::ceil(kImbalance * (double)F / n) + 1;
  const uint64_t minW = (uint64_t)std::floor((2.0 - kImbalance) * (double)F / n);
  // 3. refine, project, refine ...
  for (size_t l = levels.size(); l-- > 0;) {
    const Graph &g = levels[l];
    std::vector<uint64_t> W((size_t)n, 0);
    for (uint32_t u = 0; u < g.N; u++) W[part[u]] += g.vwgt[u];
    refine(g, part, W, maxW, minW, l == 0 ? 3 : 6);
    if (l > 0) {
      const Graph &fine = levels[l - 1];
      std::vector<uint32_t> finePart(fine.N);
      for (uint32_t u = 0; u < fine.N; u++) finePart[u] = part[fine.cmap[u]];
      part.swap(finePart);
      levels.pop_back();                                // free the coarse level
    }
  }
  if (epartOut) for (uint32_t f = 0; f < F; f++) epartOut[f] = part[f];

  // 4. node partition: the part holding most of a vertex's faces, ties to the lowest id
  const int H3 = 3 * (int)F;
  std::vector<uint64_t> count((size_t)n, 0);
  uint32_t lastPart = 0;
  for (uint32_t v = 0; v < V; v++) {
    uint32_t ps[16], cs[16];
    int np = 0;
    for (int e = m.rowptr[v]; e < m.rowptr[v + 1]; e++) {
      const int h = m.row_he[e];
      if (h >= H3) continue;
      const uint32_t p = part[h / 3];
      int k = 0;
      while (k < np && ps[k] != p) k++;
      if (k == np) { if (np == 16) continue; ps[np] = p; cs[np] = 0; np++; }
      cs[k]++;
    }
    uint32_t best = lastPart, bc = 0;                   // a vertex without faces goes with its predecessor
    for (int k = 0; k < np; k++)
      if (cs[k] > bc || (cs[k] == bc && ps[k] < best)) { best = ps[k]; bc = cs[k]; }
    npart[v] = best; count[best]++; lastPart = best;
  }
  // vertex cap: boundary vertices of a part above the cap move to the lightest adjacent part below it; parts that
  // ended up empty (possible only for tiny parts) take a vertex from their heaviest neighbour
  const uint64_t capV = (uint64_t)std::ceil(kImbalance * (double)V / n) + 1;
  for (int pass = 0; pass < 4; pass++) {
    size_t moved = 0;
    for (uint32_t v = 0; v < V; v++) {
      const uint32_t a = (uint32_t)npart[v];
      if (count[a] <= capV) continue;
      uint32_t best = 0xffffffffu;
      for (int e = m.rowptr[v]; e < m.rowptr[v + 1]; e++) {
        const uint32_t b = (uint32_t)npart[m.col[e]];
        if (b != a && count[b] + 1 <= capV && (best == 0xffffffffu || count[b] < count[best] || (count[b] == count[best] && b < best))) best = b;
      }
      if (best == 0xffffffffu) continue;
      npart[v] = best; count[a]--; count[best]++; moved++;
    }
    if (!moved) break;
  }
  for (int p = 0; p < n; p++) {
    if (count[p] > 0) continue;
    // give the empty part the highest-numbered vertex of the heaviest part (keeps every part non-empty: the
    // contract of createOrderArray, partitioning_common.cpp:112-139, assumes nothing about sizes, the kernels do)
    uint32_t heavy = (uint32_t)(std::max_element(count.begin(), count.end()) - count.begin());
    if (count[heavy] < 2) return FEM_ERR_INVALID;
    for (uint32_t v = V; v-- > 0;)
      if ((uint32_t)npart[v] == heavy) { npart[v] = p; count[heavy]--; count[p]++; break; }
  }
  return FEM_OK;
}

}  // namespace fem
