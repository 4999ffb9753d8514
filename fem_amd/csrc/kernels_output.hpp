// kernels_output.hpp -- result output (format()) and the halo pack kernel
// Included by device.hip inside its anonymous namespace (one translation unit: the kernels are
// launched from the C-ABI functions there).
#pragma once

// ---------------------------------------------------------------------------
// Result output (formatKernel + the two thrust reductions of format(), reference
// src/FEM_common.cu:183-230, without GL): min/max of n over the vertices, then
// RGBA = (n^, 1 - n^, 1 - n^, 1) with n^ = (n - min) / (max - min).
// ---------------------------------------------------------------------------
template <typename T2>
__global__ __launch_bounds__(256) void minmax_kernel(const T2 *__restrict__ state, uint V, double *__restrict__ part) {
  __shared__ double s_lo[4], s_hi[4];
  double lo = 1e300, hi = -1e300;
  for (uint i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
    const double x = (double)state[i].x;
    lo = fmin(lo, x); hi = fmax(hi, x);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = fmin(fmin(s_lo[0], s_lo[1]), fmin(s_lo[2], s_lo[3]));
    part[2 * blockIdx.x + 1] = fmax(fmax(s_hi[0], s_hi[1]), fmax(s_hi[2], s_hi[3]));
  }
}

__global__ __launch_bounds__(256) void minmax_final_kernel(const double *__restrict__ part, uint n, double *__restrict__ out) {
  __shared__ double s_lo[4], s_hi[4];
  double lo = 1e300, hi = -1e300;
  for (uint i = threadIdx.x; i < n; i += 256) { lo = fmin(lo, part[2 * i]); hi = fmax(hi, part[2 * i + 1]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = fmin(fmin(s_lo[0], s_lo[1]), fmin(s_lo[2], s_lo[3]));
    out[1] = fmax(fmax(s_hi[0], s_hi[1]), fmax(s_hi[2], s_hi[3]));
  }
}

template <typename T2>
__global__ __launch_bounds__(256) void format_kernel(const T2 *__restrict__ state, uint V, const double *__restrict__ mm,
                                                     double4 *__restrict__ rgba) {
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  // a uniform field has max == min: the reference divides by zero there (FEM_common.cu:186-193, NaN colours);
  // here it maps to n^ = 0
  const double den = mm[1] - mm[0];
  const double n = den > 0 ? ((double)state[i].x - mm[0]) / den : 0.0;
  rgba[i] = make_double4(n, 1 - n, 1 - n, 1.0);
}

// extrema of n itself (the buffers hold deviations from base_n) for a caller-owned device pair
__global__ void minmax_export_kernel(const double *__restrict__ mm, double base_n, double *__restrict__ out) {
  if (threadIdx.x < 2) out[threadIdx.x] = mm[threadIdx.x] + base_n;
}

// halo pack: dst[i] = state[send_idx[i]] (8 or 16 B per lane, coalesced stores)
template <typename T2>
__global__ __launch_bounds__(256) void pack_halo_kernel(const T2 *__restrict__ state, const uint *__restrict__ send_idx,
                                                        T2 *__restrict__ dst, uint n) {
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = state[send_idx[i]];
}

