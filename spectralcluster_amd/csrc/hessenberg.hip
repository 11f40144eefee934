// Householder reduction of a general n x n matrix to upper Hessenberg form on the device --
// the O(n^3) half of the dense general eigensolver for n > 64 (reference
// utils.py:59 calls np.linalg.eig on the whole matrix, whatever its size).  What follows the
// reduction -- the QR iteration on the Hessenberg matrix, inverse iteration for the few
// eigenvectors k-means reads, the back-transform through the reflectors -- is a serial recurrence
// per eigenvalue and runs on the host (host_eig.cpp: host_hessenberg_*).
//
// Step k (k = 0 .. n - 3) applies P_k = I - tau_k v_k v_k^T from both sides, B <- P_k B P_k,
// with v_k on rows k + 1 .. n - 1 (v_k[k + 1] = 1) chosen so that column k vanishes below the
// subdiagonal.  LAPACK dgehd2's storage: v_k[k + 2 ..] stays in the annihilated part of column k,
// tau in its own vector.  Two launches per step, both streaming (fp64, HBM / L2 bound):
//   k_hess_right   B[:, k+1:] <- B[:, k+1:] P_k     one wavefront per row: u_i = <B_i, v>, then
//                                                   B_i -= tau u_i v (the row stays in cache);
//   k_hess_left    B[k+1:, k+1:] <- P_k B[k+1:, k+1:]  one workgroup per 32 columns: w = v^T B in
//                  a first walk down its columns, the rank-1 update in a second; the workgroup
//                  that owns column k + 1 then forms reflector k + 1 from it (that column is
//                  final once both updates of step k are in), so the chain needs no third launch.
// Every sum has a fixed order: the reduction is a deterministic function of its input.
#include <hip/hip_runtime.h>

#include "sc_internal.h"

namespace sc {

namespace {

// Reflector kk from column kk of B (rows kk + 1 .. n - 1), LAPACK dlarfg: on return
// B[kk+1][kk] = beta, B[kk+2 ..][kk] = v[kk+2 ..], v[kk+1] = 1, tau[kk].  One workgroup (256).
__device__ __forceinline__ void hess_reflector(double* __restrict__ B, int ld, int n, int kk,
                                               double* __restrict__ v, double* __restrict__ tau,
                                               double* sm /* >= 4 doubles */) {
  const int tid = threadIdx.x;
  double ss = 0.0;
  for (int i = kk + 2 + tid; i < n; i += 256) {
    const double x = B[(size_t)i * ld + kk];
    ss += x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  __syncthreads();
  if ((tid & 63) == 0) sm[tid >> 6] = ss;
  __syncthreads();
  const double xnorm2 = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  const double alpha = B[(size_t)(kk + 1) * ld + kk];
  if (!(xnorm2 > 0.0)) {  // nothing to annihilate (or NaN: the host sees it in the matrix)
    if (tid == 0) {
      tau[kk] = xnorm2 == 0.0 ? 0.0 : xnorm2;
      v[kk + 1] = 1.0;
    }
    for (int i = kk + 2 + tid; i < n; i += 256) v[i] = 0.0;
    return;
  }
  const double nrm = sqrt(alpha * alpha + xnorm2);
  const double beta = alpha >= 0.0 ? -nrm : nrm;
  const double scale = 1.0 / (alpha - beta);
  for (int i = kk + 2 + tid; i < n; i += 256) {
    const double vi = B[(size_t)i * ld + kk] * scale;
    v[i] = vi;
    B[(size_t)i * ld + kk] = vi;
  }
  if (tid == 0) {
    tau[kk] = (beta - alpha) / beta;
    v[kk + 1] = 1.0;
    B[(size_t)(kk + 1) * ld + kk] = beta;
  }
}

}  // namespace

__global__ __launch_bounds__(256) void k_hess_first(double* __restrict__ B, int ld, int n,
                                                    double* __restrict__ v,
                                                    double* __restrict__ tau) {
  __shared__ double sm[4];
  hess_reflector(B, ld, n, 0, v, tau, sm);
}

// B[:, k+1:] <- B[:, k+1:] (I - tau v v^T): one wavefront per row
__global__ __launch_bounds__(256) void k_hess_right(double* __restrict__ B, int ld, int n, int k,
                                                    const double* __restrict__ v,
                                                    const double* __restrict__ tau) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const double t = tau[k];
  if (row >= n || t == 0.0) return;
  double* x = B + (size_t)row * ld;
  double u = 0.0;
  for (int j = k + 1 + lane; j < n; j += 64) u += x[j] * v[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) u += __shfl_xor(u, o);
  u *= t;
  for (int j = k + 1 + lane; j < n; j += 64) x[j] -= u * v[j];
}

// B[k+1:, k+1:] <- (I - tau v v^T) B[k+1:, k+1:]: one workgroup per 32 columns; workgroup 0 owns
// column k + 1 and leaves reflector k + 1 (vnext, tau[k + 1]) behind
__global__ __launch_bounds__(256) void k_hess_left(double* __restrict__ B, int ld, int n, int k,
                                                   const double* __restrict__ v,
                                                   double* __restrict__ tau,
                                                   double* __restrict__ vnext) {
  __shared__ double red[8][33];
  __shared__ double sm[4];
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int j = k + 1 + 32 * (int)blockIdx.x + c;
  const double t = tau[k];
  if (t != 0.0) {
    double acc = 0.0;
    if (j < n)
      for (int i = k + 1 + r; i < n; i += 8) acc += v[i] * B[(size_t)i * ld + j];
    red[r][c] = acc;
    __syncthreads();
    double w = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) w += red[q][c];
    w *= t;
    if (j < n)
      for (int i = k + 1 + r; i < n; i += 8) B[(size_t)i * ld + j] -= v[i] * w;
  }
  if (blockIdx.x == 0 && k + 1 <= n - 3) {
    __syncthreads();  // this workgroup's column k + 1 is complete and visible to all its threads
    hess_reflector(B, ld, n, k + 1, vnext, tau, sm);
  }
}

// In place: B (n x n, ld) -> Hessenberg form + reflectors (dgehd2 storage), tau (n entries),
// vwork: 2 n doubles.
void launch_hessenberg(hipStream_t s, double* B, int ld, int n, double* tau, double* vwork) {
  if (n < 3) return;
  double* vb[2] = {vwork, vwork + n};
  hipLaunchKernelGGL(k_hess_first, dim3(1), dim3(256), 0, s, B, ld, n, vb[0], tau);
  for (int k = 0; k <= n - 3; ++k) {
    const double* v = vb[k & 1];
    hipLaunchKernelGGL(k_hess_right, dim3((n + 3) / 4), dim3(256), 0, s, B, ld, n, k, v, tau);
    hipLaunchKernelGGL(k_hess_left, dim3((n - k - 1 + 31) / 32), dim3(256), 0, s, B, ld, n, k, v,
                       tau, vb[(k + 1) & 1]);
  }
}

// A <- f * A (f a power of two: exact).  The reduction forms x^T x of its columns without
// dlarfg's safmin rescaling: entries around 1e-160 underflow to a skipped reflector, around 1e155
// overflow.  gen_dense_large brings a badly scaled matrix to max|a| in [1, 2) first
// and gives the eigenvalues their factor back; a matrix within 2^+-400 of 1 is left alone.
__global__ void k_scale_matrix(double* __restrict__ A, int ld, int n, double f) {
  const int row = blockIdx.x;
  double* x = A + (size_t)row * ld;
  for (int j = threadIdx.x; j < n; j += blockDim.x) x[j] *= f;
}
void launch_scale_matrix(hipStream_t s, double* A, int ld, int n, double f) {
  hipLaunchKernelGGL(k_scale_matrix, dim3(n), dim3(256), 0, s, A, ld, n, f);
}

}  // namespace sc
