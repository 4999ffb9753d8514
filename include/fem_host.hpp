// fem_host.hpp -- the reference's C++ host interfaces for this path, on the build's own mesh type
// (C++17; no OpenMesh, no METIS).  Exported by libfem_hip.so; the flat C forms are in fem_hip.h.
//
//   SimpleTriMesh                       reference src/common.h:41-52 (OpenMesh TriMesh_ArrayKernelT)
//   MeshStats                           common.h:19-38 (hot fields)
//   initMeshStats / deleteMeshStats     mesh_function.h:124-125, mesh_function.cpp:27-67
//   computeLaplacianCPU                 mesh_function.h:6-33      (m, fn, lap, wHej, wVtx)
//   computeFaceGradientsCPU             mesh_function.h:35-59     (m, gradV12, gradV13, opGrad, fn)
//   computeVertexGradientsCPU           mesh_function.h:61-110    (m, faceGrad, opVtxGrad, meshAngles)
//   partition                           partitioning_common.h:3-9 (originalMesh, &orderedMesh, &node_parts, &element_parts, n)
// Same names and argument orders, so a caller written against the reference's headers compiles against this one
// after replacing the OpenMesh iterators it uses itself (tests/cpp/host_interface.cpp is such a caller).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "fem_hip.h"

namespace fem {

typedef unsigned int uint;

// Indexed triangle mesh + the half-edge tables the path needs.
// Conventions (DESIGN.md "Topology conventions"):
//   interior half-edge h = 3f+k runs faces[3f+k] -> faces[3f+(k+1)%3];
//   boundary half-edges are 3F+b, in ascending order of their interior
//   opposite; halfedge_handle(f) is the last one (c->a), so fv/fh iteration
//   yields a, b, c and (v1, v2, v3) = (c, a, b);
//   voh(v) = outgoing half-edges in ascending to-vertex order.
struct SimpleTriMesh {
  std::vector<double> points;          // 3V
  std::vector<uint> faces;             // 3F
  std::vector<unsigned char> colors;   // 3V or empty

  int H = 0;                           // half-edges incl. boundary; 0 = topology not built
  std::vector<int32_t> opp;            // [H]
  std::vector<uint> he_from, he_to;    // [H]
  std::vector<int32_t> rowptr;         // [V+1]
  std::vector<uint> col;               // [H]  to-vertex, ascending per row
  std::vector<int32_t> row_he;         // [H]  half-edge id of that entry

  uint n_vertices() const { return (uint)(points.size() / 3); }
  uint n_faces() const { return (uint)(faces.size() / 3); }
  uint n_halfedges() const { return (uint)H; }
  uint add_vertex(const double *p) {
    points.insert(points.end(), p, p + 3);
    H = 0;
    return n_vertices() - 1;
  }
  uint add_face(uint a, uint b, uint c) {
    faces.push_back(a); faces.push_back(b); faces.push_back(c);
    H = 0;
    return n_faces() - 1;
  }
  void clean() { points.clear(); faces.clear(); colors.clear(); H = 0; }
  bool is_boundary(int h) const { return h >= 3 * (int)n_faces(); }
  static int next_halfedge(int h) { return (h / 3) * 3 + (h % 3 + 1) % 3; }
  static int prev_halfedge(int h) { return (h / 3) * 3 + (h % 3 + 2) % 3; }
  int update_topology();               // FEM_OK / FEM_ERR_INVALID / FEM_ERR_NONMANIFOLD
};

// Hot fields of the reference's MeshStats (common.h:19-38).  gradVec12/13 are
// stored as doubles; in FEM_PREC_MIXED they hold float-rounded values.
struct MeshStats {
  int precision = -1;
  std::vector<double> meshAngle, meshCot;    // [H]
  std::vector<double> meshArea;              // [F]
  std::vector<double> wHej;                  // [H]
  std::vector<double> wVtx, areaPerVtx;      // [V]
  std::vector<double> gradVec12, gradVec13;  // [3F]
  double maxEdgeLen = 0;
};

void initMeshStats(SimpleTriMesh &ipMesh, MeshStats &mStats, int precision = FEM_PREC_MIXED);
// the same for the operator rows of the listed vertices only (a shard's local rows): entries no listed row needs stay 0
void initMeshStatsRows(SimpleTriMesh &ipMesh, MeshStats &mStats, int precision, const std::vector<uint> &rows);
void deleteMeshStats(MeshStats &mStats);

bool computeLaplacianCPU(const SimpleTriMesh &m, const double *fn, double *lap,
                         const double *wHej, const double *wVtx);
// gradients are 3 doubles per face / vertex (the reference's Vec3f arrays); precision = FEM_PREC_MIXED reproduces
// its float arithmetic on them, FEM_PREC_F64 is the all-double variant
bool computeFaceGradientsCPU(const SimpleTriMesh &m, const double *gradV12, const double *gradV13,
                             double *opGrad, const double *fn, int precision = FEM_PREC_MIXED);
bool computeVertexGradientsCPU(const SimpleTriMesh &m, const double *faceGrad, double *opVtxGrad,
                               const double *meshAngles, int precision = FEM_PREC_MIXED);

// partitioning_common.h:3-9 on the build's mesh type (same argument order): node_parts[n+1] and element_parts[n+2]
// are new[]-allocated and owned by the caller (delete[], as main.cpp:598-599 does); returns the edge cut.
int partition(const SimpleTriMesh &originalMesh, SimpleTriMesh *orderedMesh, uint **node_parts,
              uint **element_parts, int n);

// pieces of the above (partitioning_common.cpp:32-176)
int partitionVertices(const SimpleTriMesh &m, int n, int64_t *npart);          // coordinate bisection
int partitionVerticesGraph(const SimpleTriMesh &m, int n, int64_t *npart);     // graph bisection (needs topology)
int partitionVerticesAuto(const SimpleTriMesh &m, int n, int64_t *npart, int *chosen = nullptr);   // coordinate bisection unless its cut says the coordinates mislead; then graph bisection
// multilevel k-way on the dual graph (the METIS_PartMeshDual stage, partitioning_common.cpp:51-58; needs topology);
// epart (optional, [F]) receives the element partition the node partition was induced from
int partitionVerticesMultilevel(const SimpleTriMesh &m, int n, int64_t *npart, int64_t *epart = nullptr);
int64_t dualEdgeCut(const SimpleTriMesh &m, const int64_t *epart);   // dual edges between different parts
void faceParts(const SimpleTriMesh &m, const int64_t *npart, int n, int64_t *epart);
int createOrderArray(uint *parts, int64_t *part, int64_t N, int n);
void reorderMesh(const SimpleTriMesh &original, SimpleTriMesh *ordered, const int64_t *npart,
                 const int64_t *epart);
int64_t edgeCut(const SimpleTriMesh &m, const int64_t *npart);

// mesh IO + synthetic inputs
int readOFF(const std::string &path, SimpleTriMesh &m);
int readPLY(const std::string &path, SimpleTriMesh &m);
int writeOFF(const std::string &path, const SimpleTriMesh &m);
int makeIcosphere(int nu, double edge, SimpleTriMesh &m);
int makeStrip(int nx, int ny, double h, SimpleTriMesh &m);
int makeScannedLike(int nu, double edge, uint seed, double flipFraction, double noiseAmp, double highValFraction,
                    int highVal, SimpleTriMesh &m);
int subdivide(const SimpleTriMesh &in, int levels, SimpleTriMesh &out);
void initialCondition(int V, uint seed, double nMax, double *nFn, double *cFn);

}  // namespace fem
