// Constraint operators of the caller-side step before / after refinement
// (reference constraint.py:95-164): elementwise kernels.  The matrix inverse of
// ConstraintPropagation is composed from these and the fp64 MFMA GEMM in constraint_api.hip.
#include <hip/hip_runtime.h>

#include "sc_internal.h"

namespace sc {

// ---- AffinityIntegration (constraint.py:106-118): max(A, Q) or 0.5 (A + Q) ------
__global__ __launch_bounds__(256) void k_affinity_integration(const double* __restrict__ a,
                                                              const double* __restrict__ q,
                                                              double* __restrict__ out, int n,
                                                              int ld, int type) {
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= n) return;
  const size_t at = (size_t)row * ld + col;
  const double x = a[at], c = q[at];
  // np.maximum propagates NaN from either side; fmax would drop it
  out[at] = type == SC_INTEGRATION_MAX ? ((x != x || c != c) ? (x + c) : (x > c ? x : c))
                                       : 0.5 * (x + c);
}

// ---- ConstraintPropagation, step 1 (constraint.py:143-152) ----------------------
// dn_i = 1 / (sqrt(deg_i) + EPS);  P = alpha * ((dn_i A_ij) dn_j);  T0 = I + P
// (first factor of the Neumann product).  Padding columns are zero-filled so the GEMMs
// may read whole 16-wide K tiles.
__global__ __launch_bounds__(256) void k_cp_prepare(const double* __restrict__ a,
                                                    const double* __restrict__ deg,
                                                    double alpha, double* __restrict__ p,
                                                    double* __restrict__ t0, int n, int ld) {
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ld) return;
  const size_t at = (size_t)row * ld + col;
  if (col >= n) {
    p[at] = 0.0;
    t0[at] = 0.0;
    return;
  }
  const double di = 1.0 / (sqrt(deg[row]) + 1e-10);
  const double dj = 1.0 / (sqrt(deg[col]) + 1e-10);
  const double v = alpha * ((di * a[at]) * dj);
  p[at] = v;
  t0[at] = (row == col ? 1.0 : 0.0) + v;
}

// ---- ConstraintPropagation, last step (constraint.py:153-163) -------------------
// F = (1 - alpha)^2 (T Q T);  F > 0: 1 - (1 - F)(1 - A);  else: (1 + F) A
__global__ __launch_bounds__(256) void k_cp_adjust(const double* __restrict__ tqt,
                                                   const double* __restrict__ a, double scale,
                                                   double* __restrict__ out, int n, int ld) {
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= n) return;
  const size_t at = (size_t)row * ld + col;
  const double f = scale * tqt[at];
  const double x = a[at];
  // the reference evaluates both branches on masked operands and adds them; the
  // inactive branch contributes exactly 0 (1 - 1*1, or (1 + 0) * 0)
  out[at] = f > 0.0 ? (1.0 - (1.0 - f) * (1.0 - x)) + 0.0 : 0.0 + (1.0 + f) * x;
}

// ---- out = in^T (32x32 tiles through LDS), padding columns zeroed ----------------
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ in,
                                                   double* __restrict__ out, int n, int ld) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.y * 32, bj = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int gi = bj + r, gj = bi + tx;
    tile[r][tx] = (gi < n && gj < n) ? in[(size_t)gi * ld + gj] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int gi = bi + r, gj = bj + tx;
    if (gi < n && gj < ld) out[(size_t)gi * ld + gj] = gj < n ? tile[tx][r] : 0.0;
  }
}

// ---- 1 if in == in^T exactly (NaN counts as asymmetric), else 0, into *flag ------
__global__ __launch_bounds__(256) void k_symmetry_flag(const double* __restrict__ in, int n,
                                                       int ld, int* __restrict__ flag) {
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= n || col <= row) return;
  if (!(in[(size_t)row * ld + col] == in[(size_t)col * ld + row])) *flag = 0;
}

void launch_affinity_integration(hipStream_t s, const double* a, const double* q, double* out,
                                 int n, int ld, int type) {
  hipLaunchKernelGGL(k_affinity_integration, dim3((n + 255) / 256, n), dim3(256), 0, s, a, q,
                     out, n, ld, type);
}
void launch_cp_prepare(hipStream_t s, const double* a, const double* deg, double alpha,
                       double* p, double* t0, int n, int ld) {
  hipLaunchKernelGGL(k_cp_prepare, dim3((ld + 255) / 256, n), dim3(256), 0, s, a, deg, alpha,
                     p, t0, n, ld);
}
void launch_cp_adjust(hipStream_t s, const double* tqt, const double* a, double scale,
                      double* out, int n, int ld) {
  hipLaunchKernelGGL(k_cp_adjust, dim3((n + 255) / 256, n), dim3(256), 0, s, tqt, a, scale,
                     out, n, ld);
}
void launch_transpose(hipStream_t s, const double* in, double* out, int n, int ld) {
  const int t = (n + 31) / 32, tc = (ld + 31) / 32;
  hipLaunchKernelGGL(k_transpose, dim3(tc, t), dim3(256), 0, s, in, out, n, ld);
}
void launch_symmetry_flag(hipStream_t s, const double* in, int n, int ld, int* flag) {
  hipLaunchKernelGGL(k_symmetry_flag, dim3((n + 255) / 256, n), dim3(256), 0, s, in, n, ld,
                     flag);
}

}  // namespace sc
