// Size reduction before the spectral path: what the reference
// gets from sklearn's AgglomerativeClustering(metric="cosine", linkage="complete" |
// "average") (reference spectral_clusterer.py:170-199, multi_stage_clusterer.py:109-112,
// fallback_clusterer.py:108-113) and utils.get_cluster_centroids (utils.py:159-176).
//
// sklearn delegates to scipy.cluster.hierarchy.linkage: pdist(cosine), the nearest-
// neighbour-chain algorithm, a stable sort of the merges by height and a union-find
// relabelling; sklearn then cuts the tree with a heap of node ids (_hc_cut).  Here:
//   * distances: the fp64 MFMA GEMM of the affinity stage on the row-normalised embeddings,
//     d_ij = 1 - clip(cos_ij) (k_cosine_distance);
//   * nearest-neighbour chain: ONE persistent workgroup (k_ahc_nn_chain).  The chain is
//     inherently sequential (about 3 n steps); each step is a coalesced scan of one row of
//     the n x n distance matrix (first index of the minimum, the previous chain element wins
//     ties, exactly scipy's loop) or a Lance-Williams update of one row + column;
//   * the O(n log n) tree bookkeeping (sort, relabel, heap cut) runs on the host in callers_api.hip.
#include <hip/hip_runtime.h>

#include "sc_internal.h"

namespace sc {

__global__ __launch_bounds__(256) void k_cosine_distance(double* __restrict__ c, int n,
                                                         int ld) {
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= n) return;
  double v = c[(size_t)row * ld + col];
  v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);  // scipy clips the cosine to [-1, 1]
  c[(size_t)row * ld + col] = 1.0 - v;
}

constexpr int kAhcThreads = 1024;

// (value, index) minimum over the workgroup: smallest value, then smallest index
__device__ __forceinline__ void block_argmin(double& v, int& idx, double* s_val, int* s_idx) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(idx, o);
    if (ov < v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_val[wave] = v;
    s_idx[wave] = idx;
  }
  __syncthreads();
  v = s_val[0];
  idx = s_idx[0];
#pragma unroll
  for (int w = 1; w < kAhcThreads / 64; ++w) {
    if (s_val[w] < v || (s_val[w] == v && s_idx[w] < idx)) {
      v = s_val[w];
      idx = s_idx[w];
    }
  }
  __syncthreads();
}

// method 1: complete linkage (max), 2: average linkage (size-weighted mean).
// Z[k] = (x, y, height, size) in merge order, x < y slot indices (slot y keeps the cluster).
__global__ __launch_bounds__(kAhcThreads) void k_ahc_nn_chain(double* __restrict__ D, int ld,
                                                              int n, int method,
                                                              int* __restrict__ size,
                                                              int* __restrict__ chain,
                                                              double* __restrict__ Z) {
  __shared__ double s_val[kAhcThreads / 64];
  __shared__ int s_idx[kAhcThreads / 64];
  const int tid = threadIdx.x;
  const double inf = __builtin_huge_val();
  for (int i = tid; i < n; i += kAhcThreads) size[i] = 1;
  __syncthreads();
  int chain_len = 0;
  for (int k = 0; k < n - 1; ++k) {
    if (chain_len == 0) {  // first slot that is still a cluster
      double v = inf;
      int first = 0x7fffffff;
      for (int i = tid; i < n; i += kAhcThreads)
        if (size[i] > 0 && i < first) first = i;
      v = (double)first;
      block_argmin(v, first, s_val, s_idx);
      if (tid == 0) chain[0] = first;
      chain_len = 1;
      __syncthreads();
    }
    int x, y;
    double cur;
    while (true) {
      x = chain[chain_len - 1];
      const int y0 = chain_len > 1 ? chain[chain_len - 2] : -1;
      cur = y0 >= 0 ? D[(size_t)x * ld + y0] : inf;
      double best = inf;
      int besti = 0x7fffffff;
      const double* row = D + (size_t)x * ld;
      for (int i = tid; i < n; i += kAhcThreads) {
        if (size[i] > 0 && i != x) {
          const double v = row[i];
          if (v < best) {
            best = v;
            besti = i;
          }
        }
      }
      block_argmin(best, besti, s_val, s_idx);
      y = y0;
      if (best < cur) {
        cur = best;
        y = besti;
      }
      if (chain_len > 1 && y == y0) break;
      if (tid == 0) chain[chain_len] = y;
      ++chain_len;
      __syncthreads();
    }
    chain_len -= 2;
    if (x > y) {
      const int t = x;
      x = y;
      y = t;
    }
    const int nx = size[x], ny = size[y];
    __syncthreads();  // every thread has read the sizes
    if (tid == 0) {
      Z[(size_t)k * 4 + 0] = (double)x;
      Z[(size_t)k * 4 + 1] = (double)y;
      Z[(size_t)k * 4 + 2] = cur;
      Z[(size_t)k * 4 + 3] = (double)(nx + ny);
      size[x] = 0;
      size[y] = nx + ny;
    }
    const double* rx = D + (size_t)x * ld;
    double* ry = D + (size_t)y * ld;
    for (int i = tid; i < n; i += kAhcThreads) {
      if (i == x || i == y) continue;
      if (size[i] <= 0) continue;  // x and y are excluded above, the rest is unchanged
      const double dx = rx[i], dy = ry[i];
      const double nv = method == 1 ? (dx > dy ? dx : dy)
                                    : ((double)nx * dx + (double)ny * dy) / (double)(nx + ny);
      ry[i] = nv;
      D[(size_t)i * ld + y] = nv;
    }
    __syncthreads();
  }
}

// utils.get_cluster_centroids: out[c][f] = mean over members (rows in index order, the
// order np.mean(axis=0) adds them) of X[i][f]
__global__ __launch_bounds__(256) void k_cluster_centroids(const double* __restrict__ X, int ldx,
                                                           int n, int d,
                                                           const int* __restrict__ labels,
                                                           double* __restrict__ out) {
  const int c = blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= d) return;
  double acc = 0.0;
  int count = 0;
  for (int i = 0; i < n; ++i) {
    if (labels[i] == c) {
      acc += X[(size_t)i * ldx + f];
      ++count;
    }
  }
  out[(size_t)c * d + f] = acc / (double)count;  // empty cluster: 0 / 0 = NaN, as np.mean
}

void launch_cosine_distance(hipStream_t s, double* c, int n, int ld) {
  hipLaunchKernelGGL(k_cosine_distance, dim3((n + 255) / 256, n), dim3(256), 0, s, c, n, ld);
}
void launch_ahc_nn_chain(hipStream_t s, double* D, int ld, int n, int method, int* size,
                         int* chain, double* Z) {
  hipLaunchKernelGGL(k_ahc_nn_chain, dim3(1), dim3(kAhcThreads), 0, s, D, ld, n, method, size,
                     chain, Z);
}
void launch_cluster_centroids(hipStream_t s, const double* X, int ldx, int n, int d,
                              const int* labels, int k, double* out) {
  hipLaunchKernelGGL(k_cluster_centroids, dim3((d + 255) / 256, k), dim3(256), 0, s, X, ldx, n,
                     d, labels, out);
}

}  // namespace sc
