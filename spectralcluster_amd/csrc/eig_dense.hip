// Dense symmetric FULL-SPECTRUM eigenvalue path (reference utils.py:44-71 when every
// eigenvalue is consumed: max_clusters=None with a Laplacian reads w[1..n-1],
// utils.py:100-115; the ascending NormalizedDiff gap also reads np.max(w), utils.py:110).
//
//   Op = diag(p) + diag(c) S diag(c)  -- materialised once into a scratch matrix,
//   Householder tridiagonalisation  T = Q^T Op Q  (LAPACK dsytd2's recurrence), then
//   every eigenvalue of T by Sturm-sequence bisection (LAPACK dstebz's count).
//
// Why bisection and not implicit QL on the tridiagonal: QL is one serial chain of ~3 n^2
// rotations; the Sturm count is independent per eigenvalue, so n threads each run ~60
// counts of n steps -- the same eigenvalues to ulp * ||T||, in parallel.
// Eigenvectors: the few columns k-means takes normally come from the block Lanczos solver
// (eig.hip) on the untouched S.  When that solver cannot deliver them (clustered spectra:
// no convergence within its restart budget) they come from HERE, like LAPACK dstein +
// dormtr: inverse iteration on the tridiagonal form (host, O(n) per vector and iteration:
// host_tridiag_eigvectors in eig_driver.hip) and the back-transform Q z on the device
// (k_td_backtransform).  For that the reflectors are kept: v_j (v_j[j+1] = 1) in ROW j of
// the destroyed matrix, columns j+1 .. n-1 -- row j is dead once column j is eliminated and a
// row is contiguous, so the back-transform streams whole rows -- and tau_j in taus[j].
//
// Tridiagonalisation, per column j (two launches; all O(n^2) traffic is in the second):
//   k_td_column : finishes w of the previous reflector (needs v^T y), applies the pending
//                 rank-2 update to row j only, takes d_j, builds the next reflector v.
//   k_td_update : A22 <- A22 - (v' w'^T + w' v'^T) for the PREVIOUS reflector and, in the same
//                 pass over A22, y = A22 v for the NEW one (row sums complete inside one wave:
//                 no atomics, fixed summation order).
// The full symmetric trailing block is kept (not just a triangle) so every row's dot
// product is local to a wave; the two products of the rank-2 term are rounded separately and
// added, which keeps A exactly symmetric.  HBM/L2 traffic: 16 B per trailing entry per
// column, sum_j (n-j)^2 * 16 B = 16 n^3 / 3 bytes (2.9 TB at n = 8192, 46 GB at n = 2048).
#include <algorithm>

#include "sc_internal.h"

namespace sc {

// ---------------------------------------------------------------- materialise Op
__global__ __launch_bounds__(256) void k_td_materialize(
    const double* __restrict__ S, int ld, int n, const double* __restrict__ c,
    const double* __restrict__ p, double* __restrict__ M) {
  const int i = blockIdx.y;
  const double ci = c[i], pi = p[i];
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    double v = (ci * c[j]) * S[(size_t)i * ld + j];  // (c_i c_j) first: exactly symmetric
    if (i == j) v += pi;
    M[(size_t)i * ld + j] = v;
  }
}

// ---------------------------------------------------------------- block reductions
__device__ __forceinline__ double td_block_sum(double v, double* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += sm[w];  // fixed order, same in every thread
  return s;
}

// scal[0] = tau of the previous reflector (read), tau of the new one (written)
__global__ __launch_bounds__(1024) void k_td_column(
    double* __restrict__ A, int ld, int n, int j, const double* __restrict__ vprev,
    double* __restrict__ vnew, double* __restrict__ w, const double* __restrict__ y,
    double* __restrict__ d, double* __restrict__ e, double* __restrict__ scal,
    double* __restrict__ taus) {
  __shared__ double sm[16];
  __shared__ double s_wj;
  const int tid = threadIdx.x;
  const bool has_prev = j > 0;
  const double tau_p = has_prev ? scal[0] : 0.0;
  // ---- w' = tau' y - (tau'^2 / 2)(v'^T y) v'   (rows j .. n-1 of the previous reflector)
  if (has_prev) {
    double part = 0.0;
    for (int i = j + tid; i < n; i += 1024) part = __builtin_fma(vprev[i], y[i], part);
    const double dot = td_block_sum(part, sm);
    const double half = 0.5 * tau_p * tau_p * dot;
    for (int i = j + tid; i < n; i += 1024) {
      const double wi = tau_p * y[i] - half * vprev[i];
      w[i] = wi;
      if (i == j) s_wj = wi;
    }
    __syncthreads();
  }
  // ---- row j with the pending update; d_j; x = row[j+1 ..]
  const double wj = has_prev ? s_wj : 0.0;
  const double vj = has_prev ? vprev[j] : 0.0;
  double* row = A + (size_t)j * ld;
  double norm2 = 0.0;
  for (int k = j + tid; k < n; k += 1024) {
    double a = row[k];
    if (has_prev) a -= (vj * w[k] + wj * vprev[k]);  // w[k]: written by this same thread
    if (k == j) {
      d[j] = a;
    } else {
      vnew[k] = a;  // unscaled for now
      if (k > j + 1) norm2 = __builtin_fma(a, a, norm2);
    }
  }
  const int m = n - j - 1;
  if (m <= 0) {
    if (tid == 0) taus[j] = 0.0;
    return;
  }
  const double xnorm2 = td_block_sum(norm2, sm);  // (barriers also publish vnew[j + 1])
  const double alpha = vnew[j + 1];
  double tau = 0.0, scale = 0.0, beta = alpha;
  if (xnorm2 > 0.0) {
    beta = -copysign(sqrt(__builtin_fma(alpha, alpha, xnorm2)), alpha);
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  __syncthreads();  // everyone has read alpha before it is overwritten
  for (int k = j + 1 + tid; k < n; k += 1024) {
    const double vk = k == j + 1 ? 1.0 : vnew[k] * scale;
    vnew[k] = vk;
    row[k] = vk;  // kept for the back-transform (row j is dead from here on)
  }
  if (tid == 0) {
    e[j] = beta;
    scal[0] = tau;
    taus[j] = tau;
  }
}

// rows / columns j+1 .. n-1.  One wave = 4 rows; lane strides over column pairs.
__global__ __launch_bounds__(256) void k_td_update(
    double* __restrict__ A, int ld, int n, int j, const double* __restrict__ vprev,
    const double* __restrict__ w, const double* __restrict__ vnew, double* __restrict__ y,
    int has_prev) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c0 = j + 1;
  const int r0 = c0 + (blockIdx.x * 4 + wave) * 4;
  if (r0 >= n) return;
  int rows[4];
  double vi[4], wi[4], acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    rows[r] = min(r0 + r, n - 1);  // clamped duplicates recompute the same row; not stored
    vi[r] = has_prev ? vprev[rows[r]] : 0.0;
    wi[r] = has_prev ? w[rows[r]] : 0.0;
    acc[r] = 0.0;
  }
  const int kbeg = c0 & ~1;
  for (int k = kbeg + 2 * lane; k < n; k += 128) {
    const bool ok0 = k >= c0, ok1 = k + 1 < n;
    double vk0 = 0.0, vk1 = 0.0, wk0 = 0.0, wk1 = 0.0;
    if (has_prev) {
      if (ok0) { vk0 = vprev[k]; wk0 = w[k]; }
      if (ok1) { vk1 = vprev[k + 1]; wk1 = w[k + 1]; }
    }
    const double x0 = ok0 ? vnew[k] : 0.0, x1 = ok1 ? vnew[k + 1] : 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r > 0 && r0 + r >= n) break;
      double2* p = reinterpret_cast<double2*>(A + (size_t)rows[r] * ld + k);
      double2 a = *p;
      if (has_prev) {
        if (ok0) a.x -= (vi[r] * wk0 + wi[r] * vk0);
        if (ok1) a.y -= (vi[r] * wk1 + wi[r] * vk1);
        *p = a;
      }
      // outside [c0, n): x is 0, and the padding past column n may hold anything
      acc[r] = __builtin_fma(ok0 ? a.x : 0.0, x0, acc[r]);
      acc[r] = __builtin_fma(ok1 ? a.y : 0.0, x1, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o);
    if (lane == 0 && r0 + r < n) y[r0 + r] = acc[r];
  }
}

// ---------------------------------------------------------------- bisection
// bounds[0..2] = {gl, gu, pivmin}; e2[i] = e[i]^2
__global__ __launch_bounds__(1024) void k_td_bounds(const double* __restrict__ d,
                                                    const double* __restrict__ e, int n,
                                                    double* __restrict__ e2,
                                                    double* __restrict__ bounds) {
  __shared__ double smin[16], smax[16], se[16];
  const int tid = threadIdx.x;
  double lo = __builtin_huge_val(), hi = -__builtin_huge_val(), emax = 0.0;
  for (int i = tid; i < n; i += 1024) {
    const double el = i > 0 ? fabs(e[i - 1]) : 0.0;
    const double er = i < n - 1 ? fabs(e[i]) : 0.0;
    lo = fmin(lo, d[i] - el - er);
    hi = fmax(hi, d[i] + el + er);
    if (i < n - 1) {
      const double sq = e[i] * e[i];
      e2[i] = sq;
      emax = fmax(emax, sq);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o));
    hi = fmax(hi, __shfl_xor(hi, o));
    emax = fmax(emax, __shfl_xor(emax, o));
  }
  if ((tid & 63) == 0) {
    smin[tid >> 6] = lo;
    smax[tid >> 6] = hi;
    se[tid >> 6] = emax;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w2 = 1; w2 < 16; ++w2) {
      lo = fmin(lo, smin[w2]);
      hi = fmax(hi, smax[w2]);
      emax = fmax(emax, se[w2]);
    }
    const double tnorm = fmax(fabs(lo), fabs(hi));
    // far above the underflow threshold so that e2 / q stays finite (a reciprocal with
    // Newton steps is used, not an IEEE division); 1e-280 * scale is still ~1e-264 relative
    const double pivmin = 1e-280 * fmax(1.0, emax);
    const double slack = 2.1 * tnorm * 2.220446049250313e-16 * n + 2.1 * pivmin;
    bounds[0] = lo - slack;
    bounds[1] = hi + slack;
    bounds[2] = pivmin;
  }
}

__device__ __forceinline__ double td_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return r;
}

// number of eigenvalues of T that are <= x (dstebz's Sturm count)
__device__ __forceinline__ int td_count(const double* __restrict__ d,
                                        const double* __restrict__ e2, int n, double x,
                                        double pivmin) {
  double q = d[0] - x;
  if (fabs(q) < pivmin) q = -pivmin;
  int count = q <= 0.0;
  for (int i = 1; i < n; ++i) {
    q = (d[i] - x) - e2[i - 1] * td_rcp(q);
    if (fabs(q) < pivmin) q = -pivmin;
    count += q <= 0.0;
  }
  return count;
}

// thread k -> k-th smallest eigenvalue; written DESCENDING (theta[n - 1 - k])
__global__ __launch_bounds__(64) void k_td_bisect(const double* __restrict__ d,
                                                  const double* __restrict__ e2, int n,
                                                  const double* __restrict__ bounds,
                                                  double* __restrict__ theta_desc) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  const int kk = min(k, n - 1);  // keep the wave converged: loop-uniform loads of d / e2
  double lo = bounds[0], hi = bounds[1];
  const double pivmin = bounds[2];
  for (int it = 0; it < 80; ++it) {
    const double mid = 0.5 * (lo + hi);
    const bool stuck = !(mid > lo && mid < hi);
    if (__all(stuck)) break;
    const int cnt = td_count(d, e2, n, mid, pivmin);
    if (!stuck) {
      if (cnt >= kk + 1) hi = mid; else lo = mid;
    }
  }
  if (k < n) theta_desc[n - 1 - k] = 0.5 * (lo + hi);
}

// ---------------------------------------------------------------- back-transform
// Z[:, q] <- Q Z[:, q],  Q = H_0 H_1 ... H_{n-2},  H_j = I - tau_j v_j v_j^T (LAPACK dormtr,
// unblocked): one workgroup per column, the column in LDS (n <= kTdLdsRows) or in place in
// global memory, reflectors applied last to first; each is one contiguous row of A.  The
// workgroups of all columns walk the rows in step, so a row is fetched from HBM once.
constexpr int kTdLdsRows = 16384;

template <bool LDS>
__global__ __launch_bounds__(1024) void k_td_backtransform(
    const double* __restrict__ A, int ld, int n, const double* __restrict__ taus,
    double* __restrict__ Z, int ldz) {
  extern __shared__ double td_x[];
  __shared__ double sm[16];
  const int tid = threadIdx.x;
  double* zg = Z + (size_t)blockIdx.x * ldz;
  double* x = LDS ? td_x : zg;
  if (LDS) {
    for (int i = tid; i < n; i += 1024) x[i] = zg[i];
  }
  __syncthreads();
  // thread t owns x[t], x[t + 1024], ... for every reflector: no barrier is needed for x
  // itself, only the two inside the block sum
  for (int j = n - 2; j >= 0; --j) {
    const double tau = taus[j];
    if (tau == 0.0) continue;  // (uniform)
    const double* v = A + (size_t)j * ld;
    int i0 = tid;
    if (i0 < j + 1) i0 += ((j + 1 - i0 + 1023) >> 10) << 10;
    double part = 0.0;
    for (int i = i0; i < n; i += 1024) part = __builtin_fma(v[i], x[i], part);
    const double s = tau * td_block_sum(part, sm);
    for (int i = i0; i < n; i += 1024) x[i] = __builtin_fma(-s, v[i], x[i]);
  }
  __syncthreads();
  if (LDS) {
    for (int i = tid; i < n; i += 1024) zg[i] = x[i];
  }
}

// ---------------------------------------------------------------- launchers
void launch_td_materialize(hipStream_t s, const double* S, int ld, int n, const double* c,
                           const double* p, double* M) {
  hipLaunchKernelGGL(k_td_materialize, dim3(std::min(8, (n + 255) / 256), n), dim3(256), 0,
                     s, S, ld, n, c, p, M);
}

// A (n x n, ld; destroyed) -> d[0..n), e[0..n-1), taus[0..n); reflector j is left in
// A[j, j+1 .. n-1].  work: 4 n + 8 doubles.
void launch_tridiagonalize(hipStream_t s, double* A, int ld, int n, double* d, double* e,
                           double* taus, double* work) {
  double* v[2] = {work, work + n};
  double* w = work + 2 * (size_t)n;
  double* y = work + 3 * (size_t)n;
  double* scal = work + 4 * (size_t)n;
  for (int j = 0; j < n; ++j) {
    double* vnew = v[j & 1];
    const double* vprev = v[(j & 1) ^ 1];
    hipLaunchKernelGGL(k_td_column, dim3(1), dim3(1024), 0, s, A, ld, n, j, vprev, vnew, w, y,
                       d, e, scal, taus);
    const int m = n - j - 1;
    if (m > 0)
      hipLaunchKernelGGL(k_td_update, dim3((m + 15) / 16), dim3(256), 0, s, A, ld, n, j,
                         vprev, w, vnew, y, j > 0 ? 1 : 0);
  }
}

// every eigenvalue of the tridiagonal (d, e), descending, into theta_desc[0..n).
// work: n + 4 doubles.
void launch_tridiagonal_eigenvalues(hipStream_t s, const double* d, const double* e, int n,
                                    double* theta_desc, double* work) {
  double* e2 = work;
  double* bounds = work + n;
  hipLaunchKernelGGL(k_td_bounds, dim3(1), dim3(1024), 0, s, d, e, n, e2, bounds);
  hipLaunchKernelGGL(k_td_bisect, dim3((n + 63) / 64), dim3(64), 0, s, d, e2, n, bounds,
                     theta_desc);
}

// Z (column-major, `cols` columns of n, ldz) <- Q Z with the reflectors launch_tridiagonalize
// left in A / taus.
void launch_td_backtransform(hipStream_t s, const double* A, int ld, int n, const double* taus,
                             double* Z, int ldz, int cols) {
  if (cols <= 0) return;
  if (n <= kTdLdsRows) {
    const size_t lds = (size_t)n * sizeof(double);
    static bool raised = false;
    if (!raised) {  // above the 64 KB default of dynamic LDS
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_td_backtransform<true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize,
                          kTdLdsRows * (int)sizeof(double));
      raised = true;
    }
    hipLaunchKernelGGL(k_td_backtransform<true>, dim3(cols), dim3(1024), lds, s, A, ld, n, taus,
                       Z, ldz);
  } else {
    hipLaunchKernelGGL(k_td_backtransform<false>, dim3(cols), dim3(1024), 0, s, A, ld, n, taus,
                       Z, ldz);
  }
}

}  // namespace sc
