// mesh.cpp -- own mesh core: half-edge tables, .off IO, synthetic inputs.
// Stands in for what the reference gets from OpenMesh (src/common.h:41-52,
// src/main.cpp:156-167); conventions in fem_host.hpp / DESIGN.md.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "fem_host.hpp"

namespace fem {

int SimpleTriMesh::update_topology() {
  const int V = (int)n_vertices(), F = (int)n_faces(), H3 = 3 * F;
  H = 0;
  for (int i = 0; i < H3; i++)
    if (faces[i] >= (uint)V) return FEM_ERR_INVALID;

  // interior half-edges bucketed by from-vertex, each bucket sorted by to-vertex
  std::vector<int32_t> ptr(V + 2, 0);
  for (int h = 0; h < H3; h++) ptr[faces[h] + 2]++;
  for (int i = 0; i < V; i++) ptr[i + 2] += ptr[i + 1];
  std::vector<uint> to(H3);
  std::vector<int32_t> he(H3);
  for (int h = 0; h < H3; h++) {
    int f = h / 3, k = h % 3;
    uint a = faces[h], b = faces[3 * f + (k + 1) % 3];
    if (a == b) return FEM_ERR_INVALID;
    int pos = ptr[a + 1]++;
    to[pos] = b; he[pos] = h;
  }
  for (int v = 0; v < V; v++) {                       // insertion sort, rows are tiny
    for (int i = ptr[v] + 1; i < ptr[v + 1]; i++) {
      uint t = to[i]; int32_t hh = he[i]; int j = i - 1;
      while (j >= ptr[v] && to[j] > t) { to[j + 1] = to[j]; he[j + 1] = he[j]; j--; }
      to[j + 1] = t; he[j + 1] = hh;
    }
    for (int i = ptr[v] + 1; i < ptr[v + 1]; i++)
      if (to[i] == to[i - 1]) return FEM_ERR_NONMANIFOLD;
  }

  opp.assign(H3, -1);
  he_from.resize(H3); he_to.resize(H3);
  std::vector<int32_t> extra(V + 1, 0);               // boundary half-edges leaving each vertex
  int B = 0;
  for (int h = 0; h < H3; h++) {
    int f = h / 3, k = h % 3;
    uint a = faces[h], b = faces[3 * f + (k + 1) % 3];
    he_from[h] = a; he_to[h] = b;
    const uint *lo = to.data() + ptr[b], *hi = to.data() + ptr[b + 1];
    const uint *it = std::lower_bound(lo, hi, a);
    if (it != hi && *it == a) opp[h] = he[it - to.data()];
    else { extra[b]++; B++; }
  }
  H = H3 + B;
  opp.resize(H); he_from.resize(H); he_to.resize(H);
  for (int h = 0, b = 0; h < H3; h++)
    if (opp[h] < 0) {
      int hb = H3 + b++;
      he_from[hb] = he_to[h]; he_to[hb] = he_from[h];
      opp[hb] = h; opp[h] = hb;
    }

  rowptr.assign(V + 1, 0);
  for (int v = 0; v < V; v++) rowptr[v + 1] = rowptr[v] + (ptr[v + 1] - ptr[v]) + extra[v];
  col.resize(H); row_he.resize(H);
  std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
  for (int v = 0; v < V; v++)
    for (int i = ptr[v]; i < ptr[v + 1]; i++) { col[fill[v]] = to[i]; row_he[fill[v]++] = he[i]; }
  for (int hb = H3; hb < H; hb++) { uint v = he_from[hb]; col[fill[v]] = he_to[hb]; row_he[fill[v]++] = hb; }
  if (B > 0)
    for (int v = 0; v < V; v++) {
      if (!extra[v]) continue;
      for (int i = rowptr[v] + 1; i < rowptr[v + 1]; i++) {
        uint t = col[i]; int32_t hh = row_he[i]; int j = i - 1;
        while (j >= rowptr[v] && col[j] > t) { col[j + 1] = col[j]; row_he[j + 1] = row_he[j]; j--; }
        col[j + 1] = t; row_he[j + 1] = hh;
      }
      for (int i = rowptr[v] + 1; i < rowptr[v + 1]; i++)
        if (col[i] == col[i - 1]) { H = 0; return FEM_ERR_NONMANIFOLD; }
    }
  return FEM_OK;
}

// ---------------------------------------------------------------- .off IO

static bool readWholeFile(const std::string &path, std::vector<char> &buf) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) return false;
  fseek(fp, 0, SEEK_END);
  long n = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  buf.resize((size_t)n + 1);
  size_t got = fread(buf.data(), 1, (size_t)n, fp);
  fclose(fp);
  buf[got] = 0;
  return got == (size_t)n;
}

struct Cursor {
  char *p;
  void skip() {                                   // whitespace and '#' comments
    for (;;) {
      while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
      if (*p == '#') { while (*p && *p != '\n') p++; } else break;
    }
  }
  bool eol() {                                    // only blanks left on this line?
    char *q = p;
    while (*q == ' ' || *q == '\t' || *q == '\r') q++;
    return *q == '\n' || *q == 0 || *q == '#';
  }
  bool num(double &v) { skip(); char *e; v = strtod(p, &e); if (e == p) return false; p = e; return true; }
  bool integer(long &v) { skip(); char *e; v = strtol(p, &e, 10); if (e == p) return false; p = e; return true; }
};

// OFF / COFF (optionally with per-vertex "r g b [a]"), polygons fan-triangulated
// like a TriMesh import does.  Replaces OpenMesh::IO::read_mesh, main.cpp:156-167.
int readOFF(const std::string &path, SimpleTriMesh &m) {
  std::vector<char> buf;
  if (!readWholeFile(path, buf)) return FEM_ERR_IO;
  Cursor c{buf.data()};
  c.skip();
  char *w = c.p;
  while (*c.p && !isspace((unsigned char)*c.p)) c.p++;
  std::string magic(w, c.p);
  bool hasColor = magic.find('C') != std::string::npos;
  if (magic.size() < 3 || magic.compare(magic.size() - 3, 3, "OFF") != 0) return FEM_ERR_IO;
  long nv, nf, ne;
  if (!c.integer(nv) || !c.integer(nf) || !c.integer(ne) || nv < 0 || nf < 0) return FEM_ERR_IO;
  m.clean();
  m.points.resize(3 * (size_t)nv);
  if (hasColor) m.colors.resize(3 * (size_t)nv);
  for (long i = 0; i < nv; i++) {
    for (int k = 0; k < 3; k++)
      if (!c.num(m.points[3 * i + k])) return FEM_ERR_IO;
    if (hasColor) {
      double col[4] = {0, 0, 0, 0};
      int nc = 0;
      bool isFloat = false;
      while (nc < 4 && !c.eol()) {
        char *s = c.p;
        if (!c.num(col[nc])) return FEM_ERR_IO;
        for (char *q = s; q < c.p; q++) if (*q == '.') isFloat = true;
        nc++;
      }
      if (nc < 3) return FEM_ERR_IO;
      for (int k = 0; k < 3; k++) {
        double v = isFloat ? col[k] * 255.0 : col[k];
        m.colors[3 * i + k] = (unsigned char)std::min(255.0, std::max(0.0, std::floor(v + 0.5)));
      }
    }
  }
  m.faces.reserve(3 * (size_t)nf);
  std::vector<long> poly;
  for (long i = 0; i < nf; i++) {
    long n;
    if (!c.integer(n) || n < 3) return FEM_ERR_IO;
    poly.resize((size_t)n);
    for (long k = 0; k < n; k++)
      if (!c.integer(poly[k]) || poly[k] < 0 || poly[k] >= nv) return FEM_ERR_IO;
    for (long k = 1; k + 1 < n; k++) m.add_face((uint)poly[0], (uint)poly[k], (uint)poly[k + 1]);
    while (*c.p && *c.p != '\n') c.p++;          // ignore trailing face colour
  }
  return FEM_OK;
}

int writeOFF(const std::string &path, const SimpleTriMesh &m) {
  FILE *fp = fopen(path.c_str(), "w");
  if (!fp) return FEM_ERR_IO;
  const bool col = m.colors.size() == m.points.size() && !m.colors.empty();
  fprintf(fp, "%s\n%u %u 0\n", col ? "COFF" : "OFF", m.n_vertices(), m.n_faces());
  for (uint i = 0; i < m.n_vertices(); i++) {
    fprintf(fp, "%.9g %.9g %.9g", m.points[3 * i], m.points[3 * i + 1], m.points[3 * i + 2]);
    if (col) fprintf(fp, " %d %d %d", m.colors[3 * i], m.colors[3 * i + 1], m.colors[3 * i + 2]);
    fputc('\n', fp);
  }
  for (uint f = 0; f < m.n_faces(); f++)
    fprintf(fp, "3 %u %u %u\n", m.faces[3 * f], m.faces[3 * f + 1], m.faces[3 * f + 2]);
  return fclose(fp) == 0 ? FEM_OK : FEM_ERR_IO;
}

// ---------------------------------------------------------------- synthetic inputs

// Class-I geodesic icosahedron of frequency nu: V = 10 nu^2 + 2, F = 20 nu^2.
int makeIcosphere(int nu, double edge, SimpleTriMesh &m) {
  if (nu < 1 || nu > 4000 || !(edge > 0)) return FEM_ERR_INVALID;
  const double t = (1.0 + std::sqrt(5.0)) / 2.0;
  const double iv[12][3] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
                            {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
  const int ifc[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11},
                          {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                          {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9},
                          {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
  const int64_t n = nu;
  const int64_t V = 10 * n * n + 2, F = 20 * n * n;
  const int64_t perEdge = n - 1, perFace = (n - 1) * (n - 2) / 2;
  // radius for the requested mean edge: equilateral triangles of side `edge` tile 4 pi R^2
  const double R = edge * std::sqrt(std::sqrt(3.0) * (double)F / (16.0 * M_PI));

  std::map<std::pair<int, int>, int> edgeId;
  for (int f = 0; f < 20; f++)
    for (int k = 0; k < 3; k++) {
      int a = ifc[f][k], b = ifc[f][(k + 1) % 3];
      std::pair<int, int> key(std::min(a, b), std::max(a, b));
      if (!edgeId.count(key)) { int id = (int)edgeId.size(); edgeId[key] = id; }
    }
  if (edgeId.size() != 30) return FEM_ERR_INVALID;

  m.clean();
  m.points.resize(3 * (size_t)V);
  auto put = [&](int64_t id, const double *A, double wa, const double *B, double wb, const double *C, double wc) {
    double p[3], l = 0;
    for (int k = 0; k < 3; k++) { p[k] = (wa * A[k] + wb * B[k] + wc * C[k]); l += p[k] * p[k]; }
    l = R / std::sqrt(l);
    for (int k = 0; k < 3; k++) m.points[3 * id + k] = p[k] * l;
  };
  for (int i = 0; i < 12; i++) put(i, iv[i], 1, iv[i], 0, iv[i], 0);
  for (auto &e : edgeId) {
    int u = e.first.first, v = e.first.second;
    for (int64_t s = 1; s < n; s++)
      put(12 + e.second * perEdge + (s - 1), iv[u], (double)(n - s), iv[v], (double)s, iv[v], 0);
  }
  // grid point (i, j) of ico face f = ((n-i-j) A + i B + j C) / n
  auto vid = [&](int f, int64_t i, int64_t j) -> uint {
    const int A = ifc[f][0], B = ifc[f][1], C = ifc[f][2];
    const int64_t k = n - i - j;
    if (i == 0 && j == 0) return (uint)A;
    if (i == n) return (uint)B;
    if (j == n) return (uint)C;
    auto onEdge = [&](int u, int v, int64_t stepsFromU) -> uint {
      int e = edgeId[std::make_pair(std::min(u, v), std::max(u, v))];
      int64_t s = u < v ? stepsFromU : n - stepsFromU;
      return (uint)(12 + e * perEdge + (s - 1));
    };
    if (j == 0) return onEdge(A, B, i);
    if (k == 0) return onEdge(B, C, j);
    if (i == 0) return onEdge(C, A, k);
    // interior: rows j = 1..n-2, row j holds i = 1..n-1-j
    int64_t before = (j - 1) * (n - 1) - j * (j - 1) / 2;
    return (uint)(12 + 30 * perEdge + f * perFace + before + (i - 1));
  };
  for (int f = 0; f < 20; f++)
    for (int64_t j = 1; j <= n - 2; j++)
      for (int64_t i = 1; i <= n - 1 - j; i++)
        put(vid(f, i, j), iv[ifc[f][0]], (double)(n - i - j), iv[ifc[f][1]], (double)i, iv[ifc[f][2]], (double)j);

  m.faces.reserve(3 * (size_t)F);
  for (int f = 0; f < 20; f++)
    for (int64_t j = 0; j < n; j++)
      for (int64_t i = 0; i < n - j; i++) {
        m.add_face(vid(f, i, j), vid(f, i + 1, j), vid(f, i, j + 1));
        if (i + j < n - 1) m.add_face(vid(f, i + 1, j), vid(f, i + 1, j + 1), vid(f, i, j + 1));
      }
  return (int64_t)m.n_faces() == F ? FEM_OK : FEM_ERR_INVALID;
}


// "Scanned-like" irregular surface (BASELINE config 4; no scan data ships with the reference or the
// container): a geodesic sphere with seeded random edge flips (irregular valences), a fraction of
// vertices driven to valence >= highVal by flipping the edges of their 1-ring towards them
// (local fans), and seeded radial noise.  Flips act on the half-edge tables in O(1); the result is
// still a closed oriented 2-manifold.
int makeScannedLike(int nu, double edge, uint seed, double flipFraction, double noiseAmp, double highValFraction,
                    int highVal, SimpleTriMesh &m) {
  int rc = makeIcosphere(nu, edge, m);
  if (rc != FEM_OK) return rc;
  if (flipFraction < 0 || noiseAmp < 0 || noiseAmp >= 0.5 || highValFraction < 0 || highVal < 3) return FEM_ERR_INVALID;
  rc = m.update_topology();
  if (rc != FEM_OK) return rc;
  const uint V = m.n_vertices(), F = m.n_faces();
  const int H3 = 3 * (int)F;
  std::vector<uint> &tri = m.faces;
  std::vector<int32_t> opp(m.opp.begin(), m.opp.begin() + H3);
  std::vector<int32_t> vhe(V, -1);
  std::vector<uint> val(V, 0);
  for (int h = 0; h < H3; h++) { vhe[tri[h]] = h; val[tri[h]]++; }
  auto nx = [](int h) { return SimpleTriMesh::next_halfedge(h); };
  auto pv = [](int h) { return SimpleTriMesh::prev_halfedge(h); };
  uint64_t rng = seed ? seed : 1;
  auto rnd = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
  auto connected = [&](uint c, uint d) {            // is d in the 1-ring of c?  (closed surface: one loop)
    int h0 = vhe[c], h = h0;
    do {
      if (tri[nx(h)] == d) return true;
      h = nx(opp[h]);
    } while (h != h0);
    return false;
  };
  std::vector<double> &P = m.points;
  // min angle (as its cosine's complement: returns the largest cosine) and orientation of triangle
  // (x, y, z) on the sphere: normal . centroid > 0
  auto triOk = [&](uint x, uint y, uint z, double cosMax) {
    const double *px = &P[3 * (size_t)x], *py = &P[3 * (size_t)y], *pz = &P[3 * (size_t)z];
    double e[3][3];
    for (int k = 0; k < 3; k++) { e[0][k] = py[k] - px[k]; e[1][k] = pz[k] - py[k]; e[2][k] = px[k] - pz[k]; }
    const double nrm[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2],
                           e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    double out = 0;
    for (int k = 0; k < 3; k++) out += nrm[k] * (px[k] + py[k] + pz[k]);
    if (!(out > 0)) return false;
    for (int i = 0; i < 3; i++) {                   // angle between -e[i] and e[(i+1)%3]
      const double *u = e[i], *w = e[(i + 1) % 3];
      const double d = -(u[0] * w[0] + u[1] * w[1] + u[2] * w[2]);
      const double lu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      const double lw = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      if (!(d < cosMax * lu * lw)) return false;
    }
    return true;
  };
  auto flip = [&](int h, double cosMax) -> bool {   // edge a-b of faces (a,b,c), (b,a,d) becomes c-d
    const int o = opp[h];
    const int hn = nx(h), hp = pv(h), on = nx(o), op = pv(o);
    const uint a = tri[h], b = tri[hn], c = tri[hp], d = tri[op];
    if (c == d || val[a] <= 3 || val[b] <= 3 || connected(c, d)) return false;
    if (!triOk(a, d, c, cosMax) || !triOk(b, c, d, cosMax)) return false;   // no folded or needle faces
    const int e1 = opp[hn], e3 = opp[on];
    tri[hn] = d; tri[on] = c;                       // f1 = (a, d, c), f2 = (b, c, d)
    opp[h] = e3; opp[e3] = h;
    opp[o] = e1; opp[e1] = o;
    opp[hn] = on; opp[on] = hn;
    vhe[a] = h; vhe[b] = o; vhe[c] = hp; vhe[d] = op;
    val[a]--; val[b]--; val[c]++; val[d]++;
    return true;
  };
  // connectivity-only smoothing: every vertex to the centroid of its neighbours, back onto the sphere
  double R = 0;
  for (int k = 0; k < 3; k++) R += P[k] * P[k];
  R = std::sqrt(R);
  auto smooth = [&](int passes) {
    std::vector<double> acc(3 * (size_t)V);
    for (int it = 0; it < passes; it++) {
      std::fill(acc.begin(), acc.end(), 0.0);
      for (int h = 0; h < H3; h++) {                // closed surface: each neighbour seen once per vertex
        const uint a = tri[h], b = tri[nx(h)];
        for (int k = 0; k < 3; k++) acc[3 * (size_t)a + k] += P[3 * (size_t)b + k];
      }
      for (uint v = 0; v < V; v++) {
        double l = 0;
        for (int k = 0; k < 3; k++) l += acc[3 * (size_t)v + k] * acc[3 * (size_t)v + k];
        l = std::sqrt(l);
        if (l > 0)
          for (int k = 0; k < 3; k++) P[3 * (size_t)v + k] = acc[3 * (size_t)v + k] * (R / l);
      }
    }
  };
  const double cos25 = 0.90630778703665, cos8 = 0.99026806874157;
  const uint64_t nflips = (uint64_t)(flipFraction * 1.5 * (double)F);       // E = 1.5 F
  for (uint64_t i = 0; i < nflips; i++) flip((int)(rnd() % (uint32_t)H3), cos25);
  const uint64_t nhigh = (uint64_t)(highValFraction * (double)V);
  for (uint64_t i = 0; i < nhigh; i++) {
    const uint v = rnd() % V;
    // grow v's fan: sweep its ring, flipping each ring edge towards v where the geometry admits it
    for (int round = 0; round < 4 * highVal && (int)val[v] < highVal; round++) {
      int h = vhe[v];
      const int skip = (int)(rnd() % val[v]);
      for (int k = 0; k < skip; k++) h = nx(opp[h]);
      bool any = false;
      for (uint k = 0; k < val[v] && (int)val[v] < highVal; k++) {
        any |= flip(nx(h), cos8);                   // nx(h) is the edge x-y of face (v, x, y)
        h = nx(opp[h]);
      }
      if (!any) break;
    }
  }
  if (flipFraction > 0 || nhigh > 0) smooth(3);
  if (noiseAmp > 0)
    for (uint v = 0; v < V; v++) {
      const double u = (double)rnd() / 2147483648.0 * 2.0 - 1.0;
      // radial noise in units of the edge length, so the surface stays a graph over the sphere
      double r = 0;
      for (int k = 0; k < 3; k++) r += m.points[3 * (size_t)v + k] * m.points[3 * (size_t)v + k];
      r = std::sqrt(r);
      const double sc = 1.0 + noiseAmp * edge * u / r;
      for (int k = 0; k < 3; k++) m.points[3 * (size_t)v + k] *= sc;
    }
  m.H = 0;                                          // tables are stale: rebuild from the faces
  return FEM_OK;
}

// nx x ny planar strip tilted 45 degrees in y-z, vertex id = ix*ny + iy, faces
// (v, v+ny, v+ny+1), (v, v+ny+1, v+1) -- the shape of ResultantPattern.off
// (ResultantPattern.off:1-4,44,454-457).
int makeStrip(int nx, int ny, double h, SimpleTriMesh &m) {
  if (nx < 2 || ny < 2 || !(h > 0)) return FEM_ERR_INVALID;
  m.clean();
  const double s = h / std::sqrt(2.0);
  for (int ix = 0; ix < nx; ix++)
    for (int iy = 0; iy < ny; iy++) {
      double p[3] = {ix * h, iy * s, iy * s};
      m.add_vertex(p);
    }
  for (int iy = 0; iy + 1 < ny; iy++)
    for (int ix = 0; ix + 1 < nx; ix++) {
      uint v = (uint)(ix * ny + iy);
      m.add_face(v, v + ny, v + ny + 1);
      m.add_face(v, v + ny + 1, v + 1);
    }
  return FEM_OK;
}

// initializeData's IC, main.cpp:178-186; rand() -> explicit Lehmer generator.
void initialCondition(int V, uint seed, double nMax, double *nFn, double *cFn) {
  const double betaMax = (double)0.01f;             // main.cpp:50 is a float literal
  uint64_t x = seed % 2147483647u;
  if (x == 0) x = 1;
  for (int i = 0; i < V; i++) {
    x = (x * 48271u) % 2147483647u;
    double beta = -0.5 * betaMax + (double)x * betaMax / 2147483647.0;
    nFn[i] = nMax - beta;
    cFn[i] = 1 / (1 + nFn[i]);
  }
}

}  // namespace fem
