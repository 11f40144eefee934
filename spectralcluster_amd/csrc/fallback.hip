// The callers' decisions either side of the spectral path (SURVEY.md section 8f-N4;
// reference fallback_clusterer.py, naive_clusterer.py):
//   * k_naive_cluster: the online "naive" clusterer of "Speaker diarization with LSTM"
//     (naive_clusterer.py:24-105) -- inherently sequential over the embeddings, so ONE
//     workgroup walks them; the threads split the feature dimension of every cosine.
//   * k_affinity_stats_*: the single-cluster conditions that are plain reductions over the
//     resident n x n affinity (fallback_clusterer.py:137-153): min, neighbour min, std.
//   * k_gmm_pass: one EM pass (E-step + the sums of the next M-step + the log-likelihood)
//     of a 1- or 2-component 1-D Gaussian mixture over the strict upper triangle of the
//     affinity (fallback_clusterer.py:154-173, the AffinityGmmBic condition).
#include <hip/hip_runtime.h>

#include "sc_internal.h"

namespace sc {

__device__ __forceinline__ double block_sum_256(double v, double* s_red) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// state: centroids (row-major, capacity >= k0 + n), counts, *n_centroids (in/out)
__global__ __launch_bounds__(256) void k_naive_cluster(const double* __restrict__ X, int n, int d,
                                                       double threshold, double adapt_threshold,
                                                       double* __restrict__ centroids,
                                                       int* __restrict__ counts,
                                                       int* __restrict__ n_centroids,
                                                       int* __restrict__ labels) {
  __shared__ double s_red[4];
  const int tid = threadIdx.x;
  int k = *n_centroids;
  for (int i = 0; i < n; ++i) {
    const double* e = X + (size_t)i * d;
    if (k == 0) {  // first embedding: naive_clusterer.py:68-71
      for (int f = tid; f < d; f += 256) centroids[f] = e[f];
      if (tid == 0) {
        counts[0] = 1;
        labels[i] = 0;
      }
      k = 1;
      __syncthreads();
      continue;
    }
    double ee = 0.0;
    for (int f = tid; f < d; f += 256) ee += e[f] * e[f];
    const double enorm = sqrt(block_sum_256(ee, s_red));
    double best = -__builtin_huge_val();
    int label = 0;
    for (int c = 0; c < k; ++c) {
      const double* cv = centroids + (size_t)c * d;
      double dot = 0.0, cc = 0.0;
      for (int f = tid; f < d; f += 256) {
        dot += cv[f] * e[f];
        cc += cv[f] * cv[f];
      }
      dot = block_sum_256(dot, s_red);
      cc = block_sum_256(cc, s_red);
      const double sim = dot / (sqrt(cc) * enorm);  // :19-22
      if (sim > best) {  // np.argmax: first maximum
        best = sim;
        label = c;
      }
    }
    if (best < threshold) {  // new cluster, :77-80
      for (int f = tid; f < d; f += 256) centroids[(size_t)k * d + f] = e[f];
      if (tid == 0) {
        counts[k] = 1;
        labels[i] = k;
      }
      ++k;
    } else {
      if (best > adapt_threshold) {  // merge, :13-17
        const double cnt = (double)counts[label];
        double* cv = centroids + (size_t)label * d;
        for (int f = tid; f < d; f += 256) cv[f] = (cv[f] * cnt + e[f]) / (cnt + 1.0);
        __syncthreads();
        if (tid == 0) counts[label] += 1;
      }
      if (tid == 0) labels[i] = label;
    }
    __syncthreads();
  }
  if (tid == 0) *n_centroids = k;
}

// ---- reductions over the affinity ------------------------------------------------------
// partial[blk] = {min over rows, min of the first superdiagonal, sum, count}
__global__ __launch_bounds__(256) void k_affinity_stats_a(const double* __restrict__ a, int n,
                                                          int ld, double* __restrict__ partial) {
  __shared__ double s_red[4];
  __shared__ double s_min[4], s_nmin[4];
  const int row = blockIdx.x;
  const double inf = __builtin_huge_val();
  double mn = inf, sum = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double v = a[(size_t)row * ld + j];
    mn = v < mn ? v : mn;   // NaN never wins: np.min would return NaN; affinities are finite
    sum += v;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double other = __shfl_xor(mn, o);
    mn = other < mn ? other : mn;
  }
  if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = mn;
  sum = block_sum_256(sum, s_red);
  if (threadIdx.x == 0) {
    double m = s_min[0];
    for (int w = 1; w < 4; ++w) m = s_min[w] < m ? s_min[w] : m;
    partial[(size_t)row * 4 + 0] = m;
    partial[(size_t)row * 4 + 1] = row + 1 < n ? a[(size_t)row * ld + row + 1] : inf;
    partial[(size_t)row * 4 + 2] = sum;
  }
  (void)s_nmin;
}
// second pass of np.std: partial[row * 4 + 3] = sum_j (a_ij - mean)^2
__global__ __launch_bounds__(256) void k_affinity_stats_b(const double* __restrict__ a, int n,
                                                          int ld, const double* __restrict__ mean,
                                                          double* __restrict__ partial) {
  __shared__ double s_red[4];
  const int row = blockIdx.x;
  const double mu = *mean;
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double dv = a[(size_t)row * ld + j] - mu;
    acc += dv * dv;
  }
  acc = block_sum_256(acc, s_red);
  if (threadIdx.x == 0) partial[(size_t)row * 4 + 3] = acc;
}
// out = {min, neighbour min, mean, std}; stage 0 fills out[0..2], stage 1 fills out[3]
__global__ void k_affinity_stats_reduce(const double* __restrict__ partial, int n, int stage,
                                        double* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (stage == 0) {
    double mn = partial[0], nmn = partial[1], sum = 0.0;
    for (int r = 0; r < n; ++r) {
      mn = partial[(size_t)r * 4] < mn ? partial[(size_t)r * 4] : mn;
      nmn = partial[(size_t)r * 4 + 1] < nmn ? partial[(size_t)r * 4 + 1] : nmn;
      sum += partial[(size_t)r * 4 + 2];
    }
    out[0] = mn;
    out[1] = nmn;
    out[2] = sum / ((double)n * (double)n);
  } else {
    double acc = 0.0;
    for (int r = 0; r < n; ++r) acc += partial[(size_t)r * 4 + 3];
    out[3] = sqrt(acc / ((double)n * (double)n));
  }
}

// ---- 1-D Gaussian mixture over the entries a_ij, j >= i + offset ------------------------
// params: {w0, mu0, var0, w1, mu1, var1} (components = 1 uses the first three).
// mode 0: hard assignment to the nearer of centres c0 = mu0, c1 = mu1 (Lloyd step of the
//         2-means initialisation): sums = {n0, s0, q0, n1, s1, q1, inertia}
// mode 1: EM pass under `params`: responsibilities -> sums {n0, s0, q0, n1, s1, q1, loglik}
// Row blocks write partial[blk * 8 + ...]; k_gmm_reduce adds them in row order.
__global__ __launch_bounds__(256) void k_gmm_pass(const double* __restrict__ a, int n, int ld,
                                                  int offset, int components, int mode,
                                                  const double* __restrict__ params,
                                                  double* __restrict__ partial) {
  __shared__ double s_red[4];
  const int row = blockIdx.x;
  const double w0 = params[0], mu0 = params[1], v0 = params[2];
  const double w1 = params[3], mu1 = params[4], v1 = params[5];
  const double kLog2Pi = 1.8378770664093453;
  const double lw0 = log(w0) - 0.5 * (kLog2Pi + log(v0));
  const double lw1 = components > 1 ? log(w1) - 0.5 * (kLog2Pi + log(v1)) : 0.0;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = row + offset + threadIdx.x; j < n; j += 256) {
    const double y = a[(size_t)row * ld + j];
    double r0 = 1.0, r1 = 0.0, extra;
    if (mode == 0) {
      const double d0 = (y - mu0) * (y - mu0), d1 = (y - mu1) * (y - mu1);
      if (d1 < d0) {  // ties go to the first centre (argmin)
        r0 = 0.0;
        r1 = 1.0;
      }
      extra = d1 < d0 ? d1 : d0;
    } else {
      const double l0 = lw0 - 0.5 * (y - mu0) * (y - mu0) / v0;
      if (components > 1) {
        const double l1 = lw1 - 0.5 * (y - mu1) * (y - mu1) / v1;
        const double mx = l0 > l1 ? l0 : l1;
        const double lse = mx + log(exp(l0 - mx) + exp(l1 - mx));
        r0 = exp(l0 - lse);
        r1 = exp(l1 - lse);
        extra = lse;
      } else {
        extra = l0;
      }
    }
    acc[0] += r0;
    acc[1] += r0 * y;
    acc[2] += r0 * y * y;
    acc[3] += r1;
    acc[4] += r1 * y;
    acc[5] += r1 * y * y;
    acc[6] += extra;
  }
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const double v = block_sum_256(acc[q], s_red);
    if (threadIdx.x == 0) partial[(size_t)row * 8 + q] = v;
  }
}
__global__ void k_gmm_reduce(const double* __restrict__ partial, int n, double* __restrict__ sums) {
  const int q = threadIdx.x;
  if (q >= 7) return;
  double acc = 0.0;
  for (int r = 0; r < n; ++r) acc += partial[(size_t)r * 8 + q];
  sums[q] = acc;
}
// min / max of the entries used by the mixture (2-means start): out = {min, max}
__global__ __launch_bounds__(256) void k_gmm_range(const double* __restrict__ a, int n, int ld,
                                                   int offset, double* __restrict__ partial) {
  __shared__ double s_mn[4], s_mx[4];
  const int row = blockIdx.x;
  const double inf = __builtin_huge_val();
  double mn = inf, mx = -inf;
  for (int j = row + offset + threadIdx.x; j < n; j += 256) {
    const double y = a[(size_t)row * ld + j];
    mn = y < mn ? y : mn;
    mx = y > mx ? y : mx;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double a0 = __shfl_xor(mn, o), a1 = __shfl_xor(mx, o);
    mn = a0 < mn ? a0 : mn;
    mx = a1 > mx ? a1 : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    s_mn[threadIdx.x >> 6] = mn;
    s_mx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mn = s_mn[w] < mn ? s_mn[w] : mn;
      mx = s_mx[w] > mx ? s_mx[w] : mx;
    }
    mn = s_mn[0] < mn ? s_mn[0] : mn;
    mx = s_mx[0] > mx ? s_mx[0] : mx;
    partial[(size_t)row * 8 + 0] = mn;
    partial[(size_t)row * 8 + 1] = mx;
  }
}
__global__ void k_gmm_range_reduce(const double* __restrict__ partial, int n,
                                   double* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double mn = partial[0], mx = partial[1];
  for (int r = 1; r < n; ++r) {
    mn = partial[(size_t)r * 8] < mn ? partial[(size_t)r * 8] : mn;
    mx = partial[(size_t)r * 8 + 1] > mx ? partial[(size_t)r * 8 + 1] : mx;
  }
  out[0] = mn;
  out[1] = mx;
}

void launch_naive_cluster(hipStream_t s, const double* X, int n, int d, double threshold,
                          double adapt_threshold, double* centroids, int* counts,
                          int* n_centroids, int* labels) {
  hipLaunchKernelGGL(k_naive_cluster, dim3(1), dim3(256), 0, s, X, n, d, threshold,
                     adapt_threshold, centroids, counts, n_centroids, labels);
}
void launch_affinity_stats(hipStream_t s, const double* a, int n, int ld, double* partial,
                           double* out) {
  hipLaunchKernelGGL(k_affinity_stats_a, dim3(n), dim3(256), 0, s, a, n, ld, partial);
  hipLaunchKernelGGL(k_affinity_stats_reduce, dim3(1), dim3(64), 0, s, partial, n, 0, out);
  hipLaunchKernelGGL(k_affinity_stats_b, dim3(n), dim3(256), 0, s, a, n, ld, out + 2, partial);
  hipLaunchKernelGGL(k_affinity_stats_reduce, dim3(1), dim3(64), 0, s, partial, n, 1, out);
}
void launch_gmm_pass(hipStream_t s, const double* a, int n, int ld, int offset, int components,
                     int mode, const double* params, double* partial, double* sums) {
  hipLaunchKernelGGL(k_gmm_pass, dim3(n), dim3(256), 0, s, a, n, ld, offset, components, mode,
                     params, partial);
  hipLaunchKernelGGL(k_gmm_reduce, dim3(1), dim3(64), 0, s, partial, n, sums);
}
void launch_gmm_range(hipStream_t s, const double* a, int n, int ld, int offset,
                      double* partial, double* out) {
  hipLaunchKernelGGL(k_gmm_range, dim3(n), dim3(256), 0, s, a, n, ld, offset, partial);
  hipLaunchKernelGGL(k_gmm_range_reduce, dim3(1), dim3(64), 0, s, partial, n, out);
}

}  // namespace sc
