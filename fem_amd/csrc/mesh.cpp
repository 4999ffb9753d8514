// mesh.cpp -- own mesh core: half-edge tables, .off IO, synthetic inputs.
// Stands in for what the reference gets from OpenMesh (src/common.h:41-52,
// src/main.cpp:156-167); conventions in fem_host.hpp / DESIGN.md.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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
  std::atomic<int> nonManifold(0);
  parallelFor((size_t)V, 200000, [&](size_t vb, size_t ve) {     // rows are independent
    for (int v = (int)vb; v < (int)ve; v++) {                   // insertion sort, rows are tiny
      for (int i = ptr[v] + 1; i < ptr[v + 1]; i++) {
        uint t = to[i]; int32_t hh = he[i]; int j = i - 1;
        while (j >= ptr[v] && to[j] > t) { to[j + 1] = to[j]; he[j + 1] = he[j]; j--; }
        to[j + 1] = t; he[j + 1] = hh;
      }
      for (int i = ptr[v] + 1; i < ptr[v + 1]; i++)
        if (to[i] == to[i - 1]) nonManifold = 1;
    }
  });
  if (nonManifold) return FEM_ERR_NONMANIFOLD;

  opp.assign(H3, -1);
  he_from.resize(H3); he_to.resize(H3);
  parallelFor((size_t)H3, 300000, [&](size_t hb, size_t hend) {  // opposite = the entry (b -> a) of row b
    for (int h = (int)hb; h < (int)hend; h++) {
      int f = h / 3, k = h % 3;
      uint a = faces[h], b = faces[3 * f + (k + 1) % 3];
      he_from[h] = a; he_to[h] = b;
      const uint *lo = to.data() + ptr[b], *hi = to.data() + ptr[b + 1];
      const uint *it = std::lower_bound(lo, hi, a);
      if (it != hi && *it == a) opp[h] = he[it - to.data()];
    }
  });
  std::vector<int32_t> extra(V + 1, 0);               // boundary half-edges leaving each vertex
  int B = 0;
  for (int h = 0; h < H3; h++)
    if (opp[h] < 0) { extra[he_to[h]]++; B++; }
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

// Decimal -> double without strtod for the common case: up to 19 significant digits and a
// decimal exponent within +-22 convert exactly with one multiply or divide by an exact power
// of ten when the digits fit 2^53 (Clinger's fast path); anything else goes to strtod.
static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
static inline bool parseDouble(const char *&p, double &v, bool *sawPoint = nullptr) {
  const char *s = p;
  bool neg = false;
  if (*s == '-' || *s == '+') { neg = *s == '-'; s++; }
  uint64_t mant = 0;
  int digits = 0, exp10 = 0;
  const char *d0 = s;
  while (*s >= '0' && *s <= '9') { if (digits < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); digits += (mant != 0); } else exp10++; s++; }
  bool any = s != d0;
  if (*s == '.') {
    if (sawPoint) *sawPoint = true;
    s++;
    const char *f0 = s;
    while (*s >= '0' && *s <= '9') { if (digits < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); digits += (mant != 0); exp10--; } s++; }
    any = any || s != f0;
  }
  if (!any) {                                       // inf / nan spellings
    char *e;
    v = strtod(p, &e);
    if (e == p) return false;
    p = e;
    return true;
  }
  if (*s == 'e' || *s == 'E') {
    const char *e = s + 1;
    bool eneg = false;
    if (*e == '-' || *e == '+') { eneg = *e == '-'; e++; }
    if (*e >= '0' && *e <= '9') {
      int ex = 0;
      while (*e >= '0' && *e <= '9') { if (ex < 10000) ex = ex * 10 + (*e - '0'); e++; }
      exp10 += eneg ? -ex : ex;
      s = e;
      if (sawPoint) *sawPoint = true;
    }
  }
  if (mant <= (1ull << 53) && exp10 >= -22 && exp10 <= 22) {
    double x = (double)mant;
    x = exp10 < 0 ? x / kPow10[-exp10] : x * kPow10[exp10];
    v = neg ? -x : x;
  } else {
    char *e;
    v = strtod(p, &e);                              // rare: long mantissas, huge exponents, inf/nan
    if (e == p) return false;
    s = e;
  }
  p = s;
  return true;
}

// Whole file as a NUL-terminated byte range: mapped (no copy; the zero fill of the last page is
// the terminator) unless the size is a multiple of the page size, then read into memory.
struct FileBytes {
  const char *data = nullptr;
  size_t size = 0;
  void *map = nullptr;
  size_t mapLen = 0;
  std::vector<char> buf;
  bool open(const std::string &path) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return readWholeFile(path, buf) && adopt(); }
    size = (size_t)st.st_size;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    if (size == 0 || size % page == 0) { ::close(fd); return readWholeFile(path, buf) && adopt(); }
    void *q = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (q == MAP_FAILED) return readWholeFile(path, buf) && adopt();
    map = q; mapLen = size; data = (const char *)q;
    return true;
  }
  bool adopt() { data = buf.data(); size = buf.size() - 1; return true; }
  ~FileBytes() { if (map) munmap(map, mapLen); }
};

struct Cursor {
  const char *p, *end;
  void skip() {                                   // whitespace and '#' comments
    for (;;) {
      while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
      if (*p == '#') { while (*p && *p != '\n') p++; } else break;
    }
  }
  void skipBlanks() { while (*p == ' ' || *p == '\t' || *p == '\r') p++; }
  bool eol() {                                    // only blanks left on this line?
    const char *q = p;
    while (*q == ' ' || *q == '\t' || *q == '\r') q++;
    return *q == '\n' || *q == 0 || *q == '#';
  }
  bool num(double &v, bool *sawPoint = nullptr) { skip(); return parseDouble(p, v, sawPoint); }
  bool integer(long &v) {
    skip();
    const char *s = p;
    bool neg = false;
    if (*s == '-' || *s == '+') { neg = *s == '-'; s++; }
    if (*s < '0' || *s > '9') return false;
    long x = 0;
    while (*s >= '0' && *s <= '9') { x = x * 10 + (*s - '0'); s++; }
    v = neg ? -x : x;
    p = s;
    return true;
  }
};

int hostThreads() {
  const char *e = getenv("FEM_HOST_THREADS");
  int n = e ? atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  return std::max(1, std::min(n, 64));
}

// One vertex record ("x y z [r g b [a]]") or one face record ("n i0 .. in-1 [colour]"); the
// cursor stands at the record's first character and is left on the next record.
static bool parseVertex(Cursor &c, bool hasColor, double *xyz, unsigned char *rgb) {
  for (int k = 0; k < 3; k++)
    if (!c.num(xyz[k])) return false;
  if (hasColor) {
    double col[4] = {0, 0, 0, 0};
    int nc = 0;
    bool isFloat = false;
    while (nc < 4 && !c.eol()) {
      if (!c.num(col[nc], &isFloat)) return false;
      nc++;
    }
    if (nc < 3) return false;
    for (int k = 0; k < 3; k++) {
      double v = isFloat ? col[k] * 255.0 : col[k];
      rgb[k] = (unsigned char)std::min(255.0, std::max(0.0, std::floor(v + 0.5)));
    }
  }
  return true;
}

static bool parseFace(Cursor &c, long nv, std::vector<long> &poly, std::vector<uint> &tris) {
  long n;
  if (!c.integer(n) || n < 3) return false;
  poly.resize((size_t)n);
  for (long k = 0; k < n; k++)
    if (!c.integer(poly[k]) || poly[k] < 0 || poly[k] >= nv) return false;
  for (long k = 1; k + 1 < n; k++) { tris.push_back((uint)poly[0]); tris.push_back((uint)poly[k]); tris.push_back((uint)poly[k + 1]); }
  while (*c.p && *c.p != '\n') c.p++;             // ignore trailing face colour
  return true;
}

// OFF / COFF (optionally with per-vertex "r g b [a]"), polygons fan-triangulated
// like a TriMesh import does.  Replaces OpenMesh::IO::read_mesh, main.cpp:156-167.
// Built for 8-16M-vertex files: one read of the file, hand-rolled number parsing, and -- when
// the body has one record per line, as every writer produces -- the lines split over host threads
// (chunks cut at line ends; a counting pass gives every chunk its first record number, so vertex
// i lands in slot i and the faces keep file order).  Files with records spanning lines or sharing
// a line fall back to the sequential token parser.
int readOFF(const std::string &path, SimpleTriMesh &m) {
  const bool timing = getenv("FEM_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  auto lap = [&](const char *what) { if (timing) { double t = now(); fprintf(stderr, "readOFF %s %.3f s\n", what, t - t0); t0 = t; } };
  FileBytes file;
  if (!file.open(path)) return FEM_ERR_IO;
  lap("file");
  Cursor c{file.data, file.data + file.size};
  c.skip();
  const char *w = c.p;
  while (*c.p && !isspace((unsigned char)*c.p)) c.p++;
  std::string magic(w, c.p);
  bool hasColor = magic.find('C') != std::string::npos;
  if (magic.size() < 3 || magic.compare(magic.size() - 3, 3, "OFF") != 0) return FEM_ERR_IO;
  long nv, nf, ne;
  if (!c.integer(nv) || !c.integer(nf) || !c.integer(ne) || nv < 0 || nf < 0) return FEM_ERR_IO;
  if (nv > 0xfffffff0l || nf > 0xfffffff0l) return FEM_ERR_LIMIT;
  // a vertex record is at least "0 0 0\n" (6 bytes), a face record at least "3 0 1 2\n" (8): counts the file
  // cannot hold are refused before anything is sized from them
  if ((size_t)nv * 6 + (size_t)nf * 8 > file.size + 2) return FEM_ERR_IO;
  m.clean();
  m.points.resize(3 * (size_t)nv);
  if (hasColor) m.colors.resize(3 * (size_t)nv);
  c.skip();
  const char *body = c.p;
  lap("header+alloc");

  // ---- parallel path: records = non-blank, non-comment lines
  const int T = (nv + nf >= 200000) ? hostThreads() : 1;
  if (T > 1) {
    std::vector<const char *> cut(T + 1);
    cut[0] = body; cut[T] = c.end;
    for (int t = 1; t < T; t++) {
      const char *q = body + (size_t)(c.end - body) * t / T;
      while (q < c.end && *q != '\n') q++;
      cut[t] = q < c.end ? q + 1 : c.end;
    }
    auto isRecord = [](const char *q) {             // first non-blank char of the line starting at q
      while (*q == ' ' || *q == '\t' || *q == '\r') q++;
      return *q != '\n' && *q != 0 && *q != '#';
    };
    std::vector<long> count(T + 1, 0);
    {
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
          long n = 0;
          for (const char *q = cut[t]; q < cut[t + 1];) {
            n += isRecord(q);
            const char *nl = (const char *)memchr(q, '\n', (size_t)(cut[t + 1] - q));
            q = nl ? nl + 1 : cut[t + 1];
          }
          count[t + 1] = n;
        });
      for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; t++) count[t + 1] += count[t];
    lap("count lines");
    if (count[T] == nv + nf) {                      // exactly one record per line
      std::vector<std::vector<uint>> tris(T);
      std::vector<int> ok(T, 1);
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
          try {
          long rec = count[t];
          std::vector<long> poly;
          const long myFaces = count[t + 1] - std::max(count[t], nv);
          if (myFaces > 0) tris[t].reserve(3 * (size_t)myFaces + 64);
          for (const char *q = cut[t]; q < cut[t + 1];) {
            const char *nl = (const char *)memchr(q, '\n', (size_t)(cut[t + 1] - q));
            const char *next = nl ? nl + 1 : cut[t + 1];
            if (isRecord(q)) {
              Cursor lc{q, next};
              bool good;
              if (rec < nv) good = parseVertex(lc, hasColor, &m.points[3 * (size_t)rec], hasColor ? &m.colors[3 * (size_t)rec] : nullptr);
              else good = parseFace(lc, nv, poly, tris[t]);
              if (good) { lc.skipBlanks(); good = lc.p <= next && (rec >= nv || *lc.p == '\n' || *lc.p == 0 || *lc.p == '#'); }
              if (!good) { ok[t] = 0; return; }
              rec++;
            }
            q = next;
          }
          } catch (...) { ok[t] = 0; }                   // out of memory in a worker: the sequential parser decides (and reports)
        });
      for (auto &x : th) x.join();
      lap("parse");
      bool allOk = true;
      for (int t = 0; t < T; t++) allOk = allOk && ok[t];
      if (allOk) {
        size_t total = 0;
        for (auto &v : tris) total += v.size();
        m.faces.resize(total);
        std::vector<size_t> off(T + 1, 0);
        for (int t = 0; t < T; t++) off[t + 1] = off[t] + tris[t].size();
        std::vector<std::thread> cp;
        for (int t = 0; t < T; t++)
          cp.emplace_back([&, t]() { std::copy(tris[t].begin(), tris[t].end(), m.faces.begin() + off[t]); });
        for (auto &x : cp) x.join();
        lap("concat");
        return FEM_OK;
      }
      // a record ran over its line (or is malformed): let the sequential parser decide
    }
  }

  // ---- sequential token parser (records may span or share lines)
  c.p = body;
  for (long i = 0; i < nv; i++)
    if (!parseVertex(c, hasColor, &m.points[3 * (size_t)i], hasColor ? &m.colors[3 * (size_t)i] : nullptr)) return FEM_ERR_IO;
  m.faces.reserve(3 * (size_t)nf);
  std::vector<long> poly;
  for (long i = 0; i < nf; i++)
    if (!parseFace(c, nv, poly, m.faces)) return FEM_ERR_IO;
  m.H = 0;
  return FEM_OK;
}

// Stanford PLY (what scans such as Lucy/bunny ship as; OpenMesh::IO::read_mesh takes it through the same
// call, main.cpp:156-167): ascii and binary_little_endian 1.0, element vertex with x y z (any scalar
// types; red green blue read when present, other properties skipped), element face with one list
// property of vertex indices (other scalar properties skipped); polygons fan-triangulated.
namespace {
struct PlyProp { std::string name; int type = 0, listCount = -1; };   // type index into kPlySize; listCount >= 0: list
const char *const kPlyNames[] = {"char", "uchar", "short", "ushort", "int", "uint", "float", "double",
                                 "int8", "uint8", "int16", "uint16", "int32", "uint32", "float32", "float64"};
const int kPlySize[] = {1, 1, 2, 2, 4, 4, 4, 8, 1, 1, 2, 2, 4, 4, 4, 8};
int plyType(const std::string &t) {
  for (int i = 0; i < 16; i++) if (t == kPlyNames[i]) return i;
  return -1;
}
double plyBinary(const char *&p, int type) {
  const int k = type % 8;
  double v = 0;
  switch (k) {
    case 0: { int8_t x; memcpy(&x, p, 1); v = x; break; }
    case 1: { uint8_t x; memcpy(&x, p, 1); v = x; break; }
    case 2: { int16_t x; memcpy(&x, p, 2); v = x; break; }
    case 3: { uint16_t x; memcpy(&x, p, 2); v = x; break; }
    case 4: { int32_t x; memcpy(&x, p, 4); v = x; break; }
    case 5: { uint32_t x; memcpy(&x, p, 4); v = x; break; }
    case 6: { float x; memcpy(&x, p, 4); v = x; break; }
    default: { double x; memcpy(&x, p, 8); v = x; break; }
  }
  p += kPlySize[type];
  return v;
}
}  // namespace

int readPLY(const std::string &path, SimpleTriMesh &m) {
  FileBytes file;
  if (!file.open(path)) return FEM_ERR_IO;
  const char *p = file.data, *end = file.data + file.size;
  auto line = [&](std::string &out) {
    if (p >= end) return false;
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    const char *e = nl ? nl : end;
    out.assign(p, e);
    while (!out.empty() && (out.back() == '\r' || out.back() == ' ')) out.pop_back();
    p = nl ? nl + 1 : end;
    return true;
  };
  std::string l;
  if (!line(l) || l != "ply") return FEM_ERR_IO;
  bool ascii = false, haveFormat = false;
  struct Elem { std::string name; long count = 0; std::vector<PlyProp> props; };
  std::vector<Elem> elems;
  for (;;) {
    if (!line(l)) return FEM_ERR_IO;
    char a[64] = "", b[64] = "", c[64] = "", d[64] = "", e[64] = "";
    const int n = sscanf(l.c_str(), "%63s %63s %63s %63s %63s", a, b, c, d, e);
    if (n < 1) continue;
    const std::string key = a;
    if (key == "end_header") break;
    if (key == "comment" || key == "obj_info") continue;
    if (key == "format") {
      if (std::string(b) == "ascii") ascii = true;
      else if (std::string(b) != "binary_little_endian") return FEM_ERR_IO;   // big-endian files: not supported
      haveFormat = true;
    } else if (key == "element" && n >= 3) {
      Elem el; el.name = b; el.count = atol(c);
      if (el.count < 0) return FEM_ERR_IO;
      elems.push_back(el);
    } else if (key == "property" && !elems.empty()) {
      PlyProp pr;
      if (std::string(b) == "list" && n >= 5) { pr.listCount = plyType(c); pr.type = plyType(d); pr.name = e; if (pr.listCount < 0) return FEM_ERR_IO; }
      else if (n >= 3) { pr.type = plyType(b); pr.name = c; }
      else return FEM_ERR_IO;
      if (pr.type < 0) return FEM_ERR_IO;
      elems.back().props.push_back(pr);
    } else return FEM_ERR_IO;
  }
  if (!haveFormat) return FEM_ERR_IO;
  m.clean();
  Cursor cur{p, end};
  long nv = -1;
  std::vector<long> poly;
  for (const Elem &el : elems) {
    const bool isV = el.name == "vertex", isF = el.name == "face";
    int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1, il = -1;
    for (size_t k = 0; k < el.props.size(); k++) {
      const PlyProp &pr = el.props[k];
      if (isV && pr.listCount >= 0) return FEM_ERR_IO;
      if (pr.name == "x") ix = (int)k; else if (pr.name == "y") iy = (int)k; else if (pr.name == "z") iz = (int)k;
      else if (pr.name == "red") ir = (int)k; else if (pr.name == "green") ig = (int)k; else if (pr.name == "blue") ib = (int)k;
      if (isF && pr.listCount >= 0 && il < 0) il = (int)k;
    }
    if (isV) {
      if (ix < 0 || iy < 0 || iz < 0 || el.count > 0xfffffff0l) return FEM_ERR_IO;
      if ((size_t)el.count * (ascii ? 6 : 3) > (size_t)(end - cur.p) + 2) return FEM_ERR_IO;   // more vertices than the file can hold
      nv = el.count;
      m.points.resize(3 * (size_t)nv);
      const bool col = ir >= 0 && ig >= 0 && ib >= 0;
      if (col) m.colors.resize(3 * (size_t)nv);
    }
    if (isF && (il < 0 || nv < 0)) return FEM_ERR_IO;
    if (isF && (el.count < 0 || (size_t)el.count * (ascii ? 8 : 4) > (size_t)(end - cur.p) + 2)) return FEM_ERR_IO;   // likewise
    if (isF) m.faces.reserve(3 * (size_t)el.count);
    for (long i = 0; i < el.count; i++)
      for (size_t k = 0; k < el.props.size(); k++) {
        const PlyProp &pr = el.props[k];
        long cnt = 1;
        if (pr.listCount >= 0) {
          double c;
          if (ascii) { if (!cur.num(c)) return FEM_ERR_IO; }
          else { if (cur.p + kPlySize[pr.listCount] > end) return FEM_ERR_IO; c = plyBinary(cur.p, pr.listCount); }
          cnt = (long)c;
          if (cnt < 0 || cnt > 1000000) return FEM_ERR_IO;
          if (isF && (int)k == il) poly.resize((size_t)cnt);
        }
        for (long q = 0; q < cnt; q++) {
          double v;
          if (ascii) { if (!cur.num(v)) return FEM_ERR_IO; }
          else { if (cur.p + kPlySize[pr.type] > end) return FEM_ERR_IO; v = plyBinary(cur.p, pr.type); }
          if (isV) {
            const int kk = (int)k;
            if (kk == ix) m.points[3 * (size_t)i] = v; else if (kk == iy) m.points[3 * (size_t)i + 1] = v;
            else if (kk == iz) m.points[3 * (size_t)i + 2] = v;
            else if (!m.colors.empty() && (kk == ir || kk == ig || kk == ib)) {
              const double s = pr.type % 8 >= 6 ? v * 255.0 : v;          // float colours are in [0, 1]
              m.colors[3 * (size_t)i + (kk == ir ? 0 : (kk == ig ? 1 : 2))] = (unsigned char)std::min(255.0, std::max(0.0, std::floor(s + 0.5)));
            }
          } else if (isF && (int)k == il) {
            if (v < 0 || v >= (double)nv) return FEM_ERR_IO;
            poly[(size_t)q] = (long)v;
          }
        }
        if (isF && (int)k == il) {
          if (cnt < 3) return FEM_ERR_IO;
          for (long q = 1; q + 1 < cnt; q++) m.add_face((uint)poly[0], (uint)poly[(size_t)q], (uint)poly[(size_t)q + 1]);
        }
      }
  }
  if (nv < 0) return FEM_ERR_IO;
  m.H = 0;
  return FEM_OK;
}

// COFF / OFF writer (the shape of ResultantPattern.off).  Records are formatted by host threads
// into per-chunk buffers and written in order.
int writeOFF(const std::string &path, const SimpleTriMesh &m) {
  FILE *fp = fopen(path.c_str(), "w");
  if (!fp) return FEM_ERR_IO;
  const bool col = m.colors.size() == m.points.size() && !m.colors.empty();
  const size_t V = m.n_vertices(), F = m.n_faces(), N = V + F;
  fprintf(fp, "%s\n%zu %zu 0\n", col ? "COFF" : "OFF", V, F);
  const int T = N >= 200000 ? hostThreads() : 1;
  const size_t chunk = 1u << 16;                     // records per buffer
  bool ok = true;
  for (size_t r0 = 0; r0 < N && ok; r0 += chunk * T) {
    std::vector<std::string> out(T);
    std::vector<std::thread> th;
    auto work = [&](int t) {
      const size_t b = std::min(N, r0 + chunk * t), e = std::min(N, b + chunk);
      std::string &o = out[t];
      o.reserve((e - b) * (col ? 56 : 44));
      char line[160];
      for (size_t r = b; r < e; r++) {
        int n;
        if (r < V) {
          n = snprintf(line, sizeof line, "%.9g %.9g %.9g", m.points[3 * r], m.points[3 * r + 1], m.points[3 * r + 2]);
          if (col) n += snprintf(line + n, sizeof line - n, " %d %d %d", m.colors[3 * r], m.colors[3 * r + 1], m.colors[3 * r + 2]);
        } else {
          const size_t f = r - V;
          n = snprintf(line, sizeof line, "3 %u %u %u", m.faces[3 * f], m.faces[3 * f + 1], m.faces[3 * f + 2]);
        }
        line[n++] = '\n';
        o.append(line, (size_t)n);
      }
    };
    if (T == 1) work(0);
    else {
      for (int t = 0; t < T; t++) th.emplace_back(work, t);
      for (auto &x : th) x.join();
    }
    for (int t = 0; t < T && ok; t++) ok = out[t].empty() || fwrite(out[t].data(), 1, out[t].size(), fp) == out[t].size();
  }
  return (fclose(fp) == 0 && ok) ? FEM_OK : FEM_ERR_IO;
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

// Upsampler (BASELINE config 4 speaks of upsampled scans): `levels` rounds of 1-to-4 midpoint
// subdivision.  New vertices = old vertices, then one per edge in the order of its lower half-edge
// id; face (a, b, c) becomes (a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca): orientation, the
// boundary and every old vertex position are kept, colours are interpolated.
int subdivide(const SimpleTriMesh &in, int levels, SimpleTriMesh &out) {
  if (levels < 0 || levels > 12) return FEM_ERR_INVALID;
  SimpleTriMesh cur = in;
  for (int l = 0; l < levels; l++) {
    if (cur.H == 0) { const int rc = cur.update_topology(); if (rc != FEM_OK) return rc; }
    const size_t V = cur.n_vertices(), F = cur.n_faces(), H = (size_t)cur.H, E = H / 2;
    if (V + E > 0xfffffff0ull || 4 * F > 0x55555550ull) return FEM_ERR_LIMIT;
    std::vector<uint> mid(H);
    SimpleTriMesh nxt;
    nxt.points = cur.points;
    nxt.points.resize(3 * (V + E));
    const bool col = cur.colors.size() == cur.points.size() && V > 0;
    if (col) { nxt.colors = cur.colors; nxt.colors.resize(3 * (V + E)); }
    uint next = (uint)V;
    for (size_t h = 0; h < H; h++) {
      const size_t o = (size_t)cur.opp[h];
      if (o < h) { mid[h] = mid[o]; continue; }
      const uint a = cur.he_from[h], b = cur.he_to[h];
      for (int k = 0; k < 3; k++) {
        nxt.points[3 * (size_t)next + k] = 0.5 * (cur.points[3 * (size_t)a + k] + cur.points[3 * (size_t)b + k]);
        if (col) nxt.colors[3 * (size_t)next + k] = (unsigned char)((cur.colors[3 * (size_t)a + k] + cur.colors[3 * (size_t)b + k] + 1) / 2);
      }
      mid[h] = next++;
    }
    nxt.faces.resize(12 * F);
    for (size_t f = 0; f < F; f++) {
      const uint a = cur.faces[3 * f], b = cur.faces[3 * f + 1], c = cur.faces[3 * f + 2];
      const uint ab = mid[3 * f], bc = mid[3 * f + 1], ca = mid[3 * f + 2];
      const uint t[12] = {a, ab, ca, ab, b, bc, ca, bc, c, ab, bc, ca};
      std::copy(t, t + 12, nxt.faces.begin() + 12 * f);
    }
    cur = std::move(nxt);
  }
  out = std::move(cur);
  out.H = 0;
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
