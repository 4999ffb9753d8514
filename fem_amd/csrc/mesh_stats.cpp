// mesh_stats.cpp -- host operator assembly (geometry -> weights) and the CPU
// operators, on the build's own mesh type.
// Follows reference src/mesh_function.cpp:27-67 (initMeshStats), :70-160
// (angles/cot), :162-216 (areas), :219-271 (Laplacian weights), :310-350 (max
// edge), :351-397 (gradient vectors) and src/mesh_function.h:6-110 (operators).
#include <algorithm>
#include <atomic>
#include <cmath>

#include "fem_host.hpp"

namespace fem {

namespace {

// geometry scalar: float reproduces the reference's Vec3f narrowing, double is
// the all-fp64 variant
template <typename G> struct Vec3 {
  G x, y, z;
  Vec3() : x(0), y(0), z(0) {}
  Vec3(G a, G b, G c) : x(a), y(b), z(c) {}
  explicit Vec3(const double *p) : x((G)p[0]), y((G)p[1]), z((G)p[2]) {}
  Vec3 operator-(const Vec3 &o) const { return Vec3(x - o.x, y - o.y, z - o.z); }
  Vec3 operator/(G s) const { return Vec3(x / s, y / s, z / s); }
  G dot(const Vec3 &o) const { return x * o.x + y * o.y + z * o.z; }
  Vec3 cross(const Vec3 &o) const { return Vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
  G norm() const { return std::sqrt(dot(*this)); }
};

// `rows` (optional): assemble only what the operator rows of these vertices need -- the faces around them and
// the half-edges leaving them; every other entry stays 0.  The sums of a listed vertex run over the same terms
// in the same order as in the full assembly, so its row is bit-identical.  maxEdgeLen is always global.
template <typename G> void assemble(const SimpleTriMesh &m, MeshStats &s, const std::vector<uint> *rows = nullptr) {
  const int V = (int)m.n_vertices(), F = (int)m.n_faces(), H = m.H, H3 = 3 * F;
  const double *P = m.points.data();
  s.meshAngle.assign(H, 0.0); s.meshCot.assign(H, 0.0);
  s.meshArea.assign(F, 0.0); s.areaPerVtx.assign(V, 0.0);
  s.wHej.assign(H, 0.0); s.wVtx.assign(V, 0.0);
  s.gradVec12.assign(3 * (size_t)F, 0.0); s.gradVec13.assign(3 * (size_t)F, 0.0);
  std::vector<uint8_t> vNeed, fNeed;
  if (rows) {
    vNeed.assign(V, 0); fNeed.assign(F, 0);
    for (uint v : *rows) vNeed[v] = 1;
    parallelFor((size_t)F, 300000, [&](size_t fb, size_t fe) {
      for (size_t f = fb; f < fe; f++) fNeed[f] = vNeed[m.faces[3 * f]] | vNeed[m.faces[3 * f + 1]] | vNeed[m.faces[3 * f + 2]];
    });
  }
  const bool all = rows == nullptr;

  // per face: the three corner angles (angle at the to-vertex of each of its
  // half-edges), cotangents, area, hat-function gradients.  Faces write only their own slots, so
  // the loop runs on the host threads; the scatters below keep their serial order (bit-identical
  // sums whatever the thread count).
  parallelFor((size_t)F, 100000, [&](size_t fb, size_t fe) {
  for (int f = (int)fb; f < (int)fe; f++) {
    if (!all && !fNeed[f]) continue;
    const uint ia = m.faces[3 * f], ib = m.faces[3 * f + 1], ic = m.faces[3 * f + 2];
    const Vec3<G> pa(P + 3 * ia), pb(P + 3 * ib), pc(P + 3 * ic);
    const Vec3<G> corner[3] = {pa, pb, pc};
    for (int k = 0; k < 3; k++) {                       // h = 3f+k : corner k -> corner k+1
      const Vec3<G> &p1 = corner[k], &p2 = corner[(k + 1) % 3], &p3 = corner[(k + 2) % 3];
      Vec3<G> v21 = p1 - p2, v23 = p3 - p2;
      G d = v21.dot(v23 / (v21.norm() * v23.norm()));   // precedence of mesh_function.cpp:126
      double ang = std::acos(d);                        // acos(float) for G = float
      s.meshAngle[3 * f + k] = ang;
      s.meshCot[3 * f + k] = 1. / std::tan(ang);
    }
    // (v1, v2, v3) = (c, a, b)
    Vec3<G> v21 = pc - pa, v23 = pb - pa;
    double area = v21.cross(v23).norm() / 2;
    s.meshArea[f] = area;

    // face normal from the double points, narrowed, re-normalised in G
    double e1[3], e2[3];
    for (int k = 0; k < 3; k++) { e1[k] = P[3 * ic + k] - P[3 * ib + k]; e2[k] = P[3 * ia + k] - P[3 * ib + k]; }
    double nd[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    double nl = std::sqrt(nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]);
    Vec3<G> N((G)(nl != 0 ? nd[0] / nl : 0.0), (G)(nl != 0 ? nd[1] / nl : 0.0), (G)(nl != 0 ? nd[2] / nl : 0.0));
    G gl = N.norm();
    if (gl != 0) N = N / gl;
    const G twoA = (G)(2 * area);
    Vec3<G> g12 = N.cross(pc - pb) / twoA;              // N x (pt1 - pt3)
    Vec3<G> g13 = N.cross(pa - pc) / twoA;              // N x (pt2 - pt1)
    s.gradVec12[3 * f] = g12.x; s.gradVec12[3 * f + 1] = g12.y; s.gradVec12[3 * f + 2] = g12.z;
    s.gradVec13[3 * f] = g13.x; s.gradVec13[3 * f + 1] = g13.y; s.gradVec13[3 * f + 2] = g13.z;
  }
  });
  for (int f = 0; f < F; f++) {                         // lumped mass, in face order (mesh_function.cpp:205-209)
    if (!all && !fNeed[f]) continue;
    const double a3 = s.meshArea[f] / 3;
    s.areaPerVtx[m.faces[3 * f + 2]] += a3; s.areaPerVtx[m.faces[3 * f]] += a3; s.areaPerVtx[m.faces[3 * f + 1]] += a3;
  }

  // cot-Laplacian rows scaled by the inverse lumped mass
  parallelFor((size_t)H, 300000, [&](size_t hb, size_t he) {
    for (int h = (int)hb; h < (int)he; h++) {
      if (!all && !vNeed[m.he_from[h]]) continue;
      const int ho = m.opp[h];
      double wg = 0;
      if (h < H3) wg += s.meshCot[SimpleTriMesh::next_halfedge(h)];
      if (ho < H3) wg += s.meshCot[SimpleTriMesh::next_halfedge(ho)];
      s.wHej[h] = wg / (2 * s.areaPerVtx[m.he_from[h]]);
    }
  });
  for (int h = 0; h < H; h++)                                         // in half-edge order
    if (all || vNeed[m.he_from[h]]) s.wVtx[m.he_from[h]] -= s.wHej[h];

  std::vector<double> part((size_t)hostThreads() + 1, 0.0);
  std::atomic<int> slot(0);
  parallelFor((size_t)H, 300000, [&](size_t hb, size_t he) {
    double mx = 0;
    for (int h = (int)hb; h < (int)he; h++) {
      double len = (Vec3<G>(P + 3 * m.he_from[h]) - Vec3<G>(P + 3 * m.he_to[h])).norm();
      if (len > mx) mx = len;
    }
    part[(size_t)slot.fetch_add(1)] = mx;
  });
  double mx = 0;
  for (double v : part) mx = std::max(mx, v);
  s.maxEdgeLen = mx;
}

}  // namespace

// Upper bound on the spectrum of -L, L = M^-1 K the assembled Laplacian rows (wHej, wVtx): what the Chebyshev solver of
// the semi-implicit step needs (device.hip:runImexFused).  The smaller of two rigorous bounds:
//  * Gershgorin on the rows, with absolute values (obtuse triangles give negative weights): |wVtx| + sum |wHej|;
//  * element by element: K is the sum of the triangles' stiffness matrices K_t (edge weights cot / 2: the Dirichlet
//    energy of a linear function, positive semi-definite whatever the angles) and M of their lumped masses M_t = A_t / 3
//    per corner, so x'Kx = sum_t x_t'K_t x_t <= sum_t lambda_t x_t'M_t x_t <= (max_t lambda_t) x'Mx with
//    lambda_t = (3 / A_t) (w1 + w2 + w3 + sqrt(w1^2 + w2^2 + w3^2 - w1 w2 - w2 w3 - w3 w1)), the largest eigenvalue of a
//    three-vertex graph Laplacian with weights w_i = cot_i / 2.  On a regular mesh this is the exact largest eigenvalue
//    (6 / h^2) where Gershgorin gives 8 / h^2.
// Faces with zero area (a shard's stats hold only the faces its rows need) are skipped; 0 when there are no stats.
double laplacianSpectrumBound(const SimpleTriMesh &m, const MeshStats &s) {
  const size_t V = m.n_vertices(), F = m.n_faces(), H = m.he_from.size();
  if (s.precision < 0 || s.wVtx.size() != V || s.wHej.size() != H || s.meshCot.size() < 3 * F || s.meshArea.size() != F) return 0;
  std::vector<double> rowAbs(V, 0.0);
  for (size_t h = 0; h < H; h++) rowAbs[m.he_from[h]] += std::fabs(s.wHej[h]);
  double gersh = 0;
  for (size_t v = 0; v < V; v++) gersh = std::max(gersh, std::fabs(s.wVtx[v]) + rowAbs[v]);
  double elem = 0;
  for (size_t f = 0; f < F; f++) {
    const double A = s.meshArea[f];
    if (!(A > 0)) continue;
    const double w1 = 0.5 * s.meshCot[3 * f], w2 = 0.5 * s.meshCot[3 * f + 1], w3 = 0.5 * s.meshCot[3 * f + 2];
    const double lam = (3.0 / A) * (w1 + w2 + w3 + std::sqrt(std::max(0.0, w1 * w1 + w2 * w2 + w3 * w3 - w1 * w2 - w2 * w3 - w3 * w1)));
    if (!std::isfinite(lam)) return gersh;
    elem = std::max(elem, lam);
  }
  return elem > 0 ? std::min(gersh, elem) : gersh;
}

void initMeshStats(SimpleTriMesh &ipMesh, MeshStats &mStats, int precision) {
  if (ipMesh.H == 0) ipMesh.update_topology();
  mStats.precision = precision;
  if (precision == FEM_PREC_F64) assemble<double>(ipMesh, mStats);
  else assemble<float>(ipMesh, mStats);
}

void initMeshStatsRows(SimpleTriMesh &ipMesh, MeshStats &mStats, int precision, const std::vector<uint> &rows) {
  if (ipMesh.H == 0) ipMesh.update_topology();
  mStats.precision = precision;
  if (precision == FEM_PREC_F64) assemble<double>(ipMesh, mStats, &rows);
  else assemble<float>(ipMesh, mStats, &rows);
}

void deleteMeshStats(MeshStats &mStats) { mStats = MeshStats(); }

bool computeLaplacianCPU(const SimpleTriMesh &m, const double *fn, double *lap, const double *wHej,
                         const double *wVtx) {
  const int V = (int)m.n_vertices();
  for (int i = 0; i < V; i++) {
    double acc = fn[i] * wVtx[i];
    for (int e = m.rowptr[i]; e < m.rowptr[i + 1]; e++) acc += fn[m.col[e]] * wHej[m.row_he[e]];
    lap[i] = acc;
  }
  return true;
}

template <typename G>
static void faceGradients(const SimpleTriMesh &m, const double *g12, const double *g13, double *out,
                          const double *fn) {
  const int F = (int)m.n_faces();
  for (int f = 0; f < F; f++) {
    const uint v1 = m.faces[3 * f + 2], v2 = m.faces[3 * f], v3 = m.faces[3 * f + 1];
    const G d12 = (G)(fn[v2] - fn[v1]), d13 = (G)(fn[v3] - fn[v1]);
    for (int k = 0; k < 3; k++) out[3 * f + k] = (G)g12[3 * f + k] * d12 + (G)g13[3 * f + k] * d13;
  }
}

bool computeFaceGradientsCPU(const SimpleTriMesh &m, const double *gradV12, const double *gradV13,
                             double *opGrad, const double *fn, int precision) {
  if (precision == FEM_PREC_F64) faceGradients<double>(m, gradV12, gradV13, opGrad, fn);
  else faceGradients<float>(m, gradV12, gradV13, opGrad, fn);
  return true;
}

template <typename G>
static void vertexGradients(const SimpleTriMesh &m, const double *fg, double *out, const double *ang) {
  const int V = (int)m.n_vertices(), H3 = 3 * (int)m.n_faces();
  for (int v = 0; v < V; v++) {
    G g[3] = {0, 0, 0};
    double wg = 0;
    for (int e = m.rowptr[v]; e < m.rowptr[v + 1]; e++) {
      const int h = m.row_he[e];
      if (h >= H3) continue;
      const int hp = SimpleTriMesh::prev_halfedge(h);
      const G a = (G)ang[hp];
      for (int k = 0; k < 3; k++) g[k] += (G)fg[3 * (h / 3) + k] * a;
      wg += ang[hp];
    }
    for (int k = 0; k < 3; k++) out[3 * v + k] = wg > 0 ? (double)(g[k] / (G)wg) : 0.0;
  }
}

bool computeVertexGradientsCPU(const SimpleTriMesh &m, const double *faceGrad, double *opVtxGrad,
                               const double *meshAngles, int precision) {
  if (precision == FEM_PREC_F64) vertexGradients<double>(m, faceGrad, opVtxGrad, meshAngles);
  else vertexGradients<float>(m, faceGrad, opVtxGrad, meshAngles);
  return true;
}

}  // namespace fem
