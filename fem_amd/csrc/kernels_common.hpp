// kernels_common.hpp -- vector types, model constants, the reaction update and streaming loads shared by the step kernels
// Included by device.hip inside its anonymous namespace (one translation unit: the kernels are
// launched from the C-ABI functions there).
#pragma once

template <typename T> struct Vec;
template <> struct Vec<float> { typedef float2 t2; typedef float4 t4; };
template <> struct Vec<double> { typedef double2 t2; typedef double4 t4; };

// bn = base value the stored n is a deviation from; cden = 1/(1+bn); nMaxOff = nMax - bn
template <typename T> struct ModelT { T nMaxOff, D, r, alpha, S, dt, bn, cden; };

// Reaction + explicit Euler + clamp (FEM_common.cu:153-165) on deviation-form state.
// The buffers hold deviations (n - bn, c - bn/(1+bn)) from the homogeneous state: differences
// are unchanged, n(nMax - n) = n((nMax - bn) - dev) and n/(1+n) - c = dev_n/((1+n)(1+bn)) - dev_c
// are exact rewrites, and the update no longer rounds at the magnitude of n itself.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double fast_div(double a, double b) { return a / b; }
template <typename T>
__device__ __forceinline__ typename Vec<T>::t2 reaction_update(typename Vec<T>::t2 s, T lapn, T lapc, T gnx, T gny,
                                                               T gnz, T gcx, T gcy, T gcz, const ModelT<T> &m) {
  const T n = m.bn + s.x;
  const T dotP = gnx * gcx + gny * gcy + gnz * gcz;
  const T dauN = m.D * lapn - m.alpha * n * lapc - m.alpha * dotP + m.S * m.r * n * (m.nMaxOff - s.x);
  const T dauC = lapc + m.S * (fast_div(s.x * m.cden, 1 + n) - s.y);
  const T un = m.dt * dauN + s.x, uc = m.dt * dauC + s.y;
  typename Vec<T>::t2 o;
  o.x = un + m.bn > 0 ? un : -m.bn;                 // max(0, n) of FEM_common.cu:164-165
  o.y = uc + m.bn * m.cden > 0 ? uc : -(m.bn * m.cden);
  return o;
}

// IMEX variant (semi-implicit stepping, kernels_implicit.hpp): the reaction and chemotaxis terms
// explicitly, the two diffusion terms (D lap n, lap c) left to the implicit solve; no clamp here.
template <typename T>
__device__ __forceinline__ typename Vec<T>::t2 imex_rhs_update(typename Vec<T>::t2 s, T lapc, T gnx, T gny, T gnz, T gcx,
                                                               T gcy, T gcz, const ModelT<T> &m) {
  const T n = m.bn + s.x;
  const T dotP = gnx * gcx + gny * gcy + gnz * gcz;
  const T dauN = -m.alpha * n * lapc - m.alpha * dotP + m.S * m.r * n * (m.nMaxOff - s.x);
  const T dauC = m.S * (s.x * m.cden / (1 + n) - s.y);
  typename Vec<T>::t2 o;
  o.x = m.dt * dauN + s.x;
  o.y = m.dt * dauC + s.y;
  return o;
}

// streaming (read-once) 16/32-byte load: keeps the operator rows from evicting the state lines in L2
__device__ __forceinline__ float4 ntload(const float4 *p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double4 ntload(const double4 *p) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d a = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p) + 1);
  return make_double4(a.x, a.y, b.x, b.y);
}

