// A caller written the way the reference's main.cpp uses its host headers (initializeData, main.cpp:169-192, and
// CPUrun, main.cpp:619-676), compiled with plain g++ against include/fem_host.hpp and linked with libfem_hip.so:
// partition(originalMesh, &orderedMesh, &node_parts, &element_parts, n), initMeshStats(mesh, mStats), the three
// compute*CPU operators with the reference's argument orders, deleteMeshStats.  Prints values the Python side of
// the test (tests/test_host_cpp_interface.py) compares with the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fem_host.hpp"

using namespace fem;

int main(int argc, char **argv) {
  const int nu = argc > 1 ? std::atoi(argv[1]) : 8, n = argc > 2 ? std::atoi(argv[2]) : 4;
  SimpleTriMesh originalMesh, orderedMesh;
  if (makeIcosphere(nu, 0.1, originalMesh) != FEM_OK || originalMesh.update_topology() != FEM_OK) return 1;
  uint *node_parts = nullptr, *element_parts = nullptr;
  const int cut = partition(originalMesh, &orderedMesh, &node_parts, &element_parts, n);      // main.cpp:170
  if (cut < 0) return 2;
  MeshStats mStats;
  initMeshStats(orderedMesh, mStats);                                                          // main.cpp:172
  const int V = (int)orderedMesh.n_vertices(), F = (int)orderedMesh.n_faces();
  std::vector<double> nFn(V), cFn(V), lapN(V), lapC(V), faceG(3 * (size_t)F), gradN(3 * (size_t)V), gradC(3 * (size_t)V);
  initialCondition(V, 1, 1.0, nFn.data(), cFn.data());
  const double dt = 0.1 * mStats.maxEdgeLen * mStats.maxEdgeLen;                               // main.cpp:188
  // one step of CPUrun (main.cpp:633-665) with the kernel's constants
  const double D = 0.25, r = 59.453790914308100, alpha = 269, S = 1, nMax = 1;
  computeLaplacianCPU(orderedMesh, nFn.data(), lapN.data(), mStats.wHej.data(), mStats.wVtx.data());
  computeLaplacianCPU(orderedMesh, cFn.data(), lapC.data(), mStats.wHej.data(), mStats.wVtx.data());
  computeFaceGradientsCPU(orderedMesh, mStats.gradVec12.data(), mStats.gradVec13.data(), faceG.data(), nFn.data());
  computeVertexGradientsCPU(orderedMesh, faceG.data(), gradN.data(), mStats.meshAngle.data());
  computeFaceGradientsCPU(orderedMesh, mStats.gradVec12.data(), mStats.gradVec13.data(), faceG.data(), cFn.data());
  computeVertexGradientsCPU(orderedMesh, faceG.data(), gradC.data(), mStats.meshAngle.data());
  for (int j = 0; j < V; j++) {
    const float dot = (float)gradC[3 * j] * (float)gradN[3 * j] + (float)gradC[3 * j + 1] * (float)gradN[3 * j + 1] +
                      (float)gradC[3 * j + 2] * (float)gradN[3 * j + 2];
    const double dn = D * lapN[j] - alpha * nFn[j] * lapC[j] - alpha * dot + S * r * nFn[j] * (nMax - nFn[j]);
    const double dc = lapC[j] + S * (-cFn[j] + nFn[j] / (1 + nFn[j]));
    nFn[j] = std::fmax(0.0, nFn[j] + dt * dn);
    cFn[j] = std::fmax(0.0, cFn[j] + dt * dc);
  }
  std::printf("V %d F %d cut %d parts", V, F, cut);
  for (int p = 0; p <= n; p++) std::printf(" %u", node_parts[p]);
  std::printf(" | elements");
  for (int p = 0; p <= n + 1; p++) std::printf(" %u", element_parts[p]);
  std::printf("\nmaxEdgeLen %.17g\n", mStats.maxEdgeLen);
  for (int j = 0; j < V; j++) std::printf("%.17g %.17g\n", nFn[j], cFn[j]);
  deleteMeshStats(mStats);
  delete[] node_parts;                                                                         // main.cpp:598-599
  delete[] element_parts;
  return mStats.wHej.empty() ? 0 : 3;
}
