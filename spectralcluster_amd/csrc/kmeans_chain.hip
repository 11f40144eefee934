// K1/K2 as a CHAIN of short multi-workgroup kernels (reference
// custom_distance_kmeans.py:13-141, custom_dist="cosine"; same algorithm and arithmetic as
// kmeans.hip, which states it in one single-workgroup kernel).
//
// Why a chain: the single-workgroup kernel streams the (n, k) embedding ~35 times through
// one CU (one pass per k-means++ trial pair, per assignment, per update) and is bound by that
// CU's L2 bandwidth (0.57-0.71 ms at n = 8192, k = 8).  Every phase here is one launch of
// n / 128 workgroups, one row per thread, that re-reads its 128 rows (1 KB per column, L2) and
// leaves a handful of per-workgroup partial sums; the NEXT launch's prologue adds the partials
// of all workgroups in workgroup order (the launch-boundary reduce: a dependent kernel
// boundary costs ~1.5 us, cheaper than any in-kernel cross-workgroup hand-off).  All sums
// are in a fixed order: deterministic.
//
//   km_colsum            column sums                          -> mean
//   km_first             |x - mean|^2, distances to the first centre (closest), pot
//   per k-means++ round: km_select  (apply the previous round's winner to `closest`, cumulative
//                                    sum in row order, searchsorted of the scaled RandomState
//                                    doubles -> candidate rows)
//                        km_trials  (potential of every candidate)
//   km_lloyd             last winner; one Euclidean Lloyd assignment on the centred data +
//                        per-cluster partial sums
//   km_cosine(it)        test of iteration it - 1 (mean-distance stop rule, :131-133), centroid
//                        update (Lloyd rule for it = 0, the `.any()`-on-indices rule after,
//                        :136-140), cosine assignment it, labels, partial sums.  Iterations are
//                        enqueued four at a time; a `done` word, read with the labels, says
//                        whether more are needed.
#include <algorithm>
#include <cstring>
#include <mutex>

#include "sc_internal.h"

namespace sc {

// threads per workgroup = rows per workgroup (128: every link's serial part -- the per-cluster
// sums over the workgroup's rows, the scan -- is half as long as with 256; k-means stage at
// n = 1650: 0.126 -> 0.107 ms, n = 8192: unchanged)
constexpr int kKmT = 128;
constexpr int kKmW = kKmT / 64;
constexpr int kKmMaxG = 1024;  // workgroups (n <= 262144) the chain is used for

struct KmChain {
  const double* ET;
  int lde, n, k, G;
  const double* rnd;
  int trials, first_center, max_iter;
  double* closest;
  double* xsq;
  long long* labels64;
  double* centroids_out;
  double* pm;     // [G][64] column sums
  double* meang;  // [64] column means (km_meanreduce)
  double* ppot;   // [G]
  double* pT;     // [G][8]
  double* pS[2];  // [G][k*k + 2k]
  double* pD[2];  // [G]
  double* meand;  // [512]
  double* cent[2];
  int* seeds;     // [64]
  int* cand[2];   // [8]
  int* info;      // [0] iterations, [8] done
};

__device__ __forceinline__ double km_wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// sum over the workgroup, same value in every thread; sm: kKmW doubles
__device__ __forceinline__ double km_bsum(double v, double* sm) {
  v = km_wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < kKmW; ++w) t += sm[w];
  return t;
}

// Sum of src[g * stride], g = 0 .. G-1, in that order.  The partials were written by
// workgroups on other XCDs, so every load is a miss (~1 us): they are issued in independent
// batches of 16 and then added in order -- a few miss latencies instead of G.
__device__ __forceinline__ double km_ordered_sum(const double* src, int stride, int G) {
  double acc = 0.0;
  for (int g0 = 0; g0 < G; g0 += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (g0 + u < G) ? src[(size_t)(g0 + u) * stride] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  return acc;
}

// every thread gets the column means km_meanreduce left
__device__ __forceinline__ void km_mean(const KmChain& a, double* mean) {
  if ((int)threadIdx.x < a.k) mean[threadIdx.x] = a.meang[threadIdx.x];
}

// mean[j] = (sum over workgroups, in order, of the column sums) / n; one workgroup
__device__ __forceinline__ void km_meanreduce_body(const KmChain& a) {
  if ((int)threadIdx.x < a.k)
    a.meang[threadIdx.x] = km_ordered_sum(a.pm + threadIdx.x, 64, a.G) / (double)a.n;
}
__global__ __launch_bounds__(64) void km_meanreduce(const KmChain a) { km_meanreduce_body(a); }
// Grouped forms (batch_group.hip): blockIdx.y picks one of up to kGroupMax independent
// problems whose argument blocks travel in the kernel arguments; G = 0 marks an idle member.
__global__ __launch_bounds__(64) void km_meanreduce_g(const GroupOf<KmChain> g) {
  const KmChain& a = g.s[blockIdx.y];
  if (a.G > 0) km_meanreduce_body(a);
}

template <int KC>
__device__ __forceinline__ void km_load_row(const KmChain& a, int r, double (&v)[KC]) {
#pragma unroll
  for (int j = 0; j < KC; ++j) v[j] = (j < a.k && r < a.n) ? a.ET[(size_t)j * a.lde + r] : 0.0;
}

template <int KC>
__device__ __forceinline__ void km_colsum_body(const KmChain& a) {
  __shared__ double sm[kKmW][KC];
  const int tid = threadIdx.x, r = blockIdx.x * kKmT + tid;
  double v[KC];
  km_load_row<KC>(a, r, v);
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    const double s = km_wsum(v[j]);
    if ((tid & 63) == 0) sm[tid >> 6][j] = s;
  }
  __syncthreads();
  if (tid < a.k) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kKmW; ++w) t += sm[w][tid];
    a.pm[(size_t)blockIdx.x * 64 + tid] = t;
  }
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_colsum(const KmChain a) {
  km_colsum_body<KC>(a);
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_colsum_g(const GroupOf<KmChain> g) {
  const KmChain& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.G) return;
  km_colsum_body<KC>(a);
}

// squared distance of this thread's (centred) row to a centred candidate row, exactly as
// sklearn's _euclidean_distances spells it: -2 x.c + |c|^2 + |x|^2, clamped at 0
template <int KC>
__device__ __forceinline__ double km_dist(const double (&v)[KC], const double* mean,
                                          const double* crow, double csq, double xsq, int k) {
  double dot = 0.0;
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (j < k) dot += crow[j] * (v[j] - mean[j]);
  double d = -2.0 * dot;
  d += csq;
  d += xsq;
  return fmax(d, 0.0);
}

template <int KC>
__device__ __forceinline__ void km_first_body(const KmChain& a) {
  __shared__ double mean[KC], crow[KC], sm[kKmW];
  __shared__ double csq;
  const int tid = threadIdx.x, r = blockIdx.x * kKmT + tid;
  km_mean(a, mean);
  double v[KC];
  km_load_row<KC>(a, r, v);
  __syncthreads();
  if (tid < a.k) crow[tid] = a.ET[(size_t)tid * a.lde + a.first_center] - mean[tid];
  __syncthreads();
  if (tid == 0) {
    double s2 = 0.0;
    for (int j = 0; j < a.k; ++j) s2 += crow[j] * crow[j];
    csq = s2;
  }
  double xs = 0.0;
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (j < a.k) {
      const double x = v[j] - mean[j];
      xs += x * x;
    }
  __syncthreads();
  double d = 0.0;
  if (r < a.n) {
    d = km_dist<KC>(v, mean, crow, csq, xs, a.k);
    a.xsq[r] = xs;
    a.closest[r] = d;
  }
  const double tot = km_bsum(d, sm);
  if (tid == 0) a.ppot[blockIdx.x] = tot;
  if (blockIdx.x == 0) {
    if (tid == 0) {
      a.seeds[0] = a.first_center;
      a.info[0] = 0;
      a.info[8] = 0;
    }
    if (tid < 8) a.cand[1][tid] = a.n - 1;  // np.clip(candidate_ids, None, n - 1)
  }
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_first(const KmChain a) {
  km_first_body<KC>(a);
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_first_g(const GroupOf<KmChain> g) {
  const KmChain& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.G) return;
  km_first_body<KC>(a);
}

// first index of the minimum of pots[0..trials) (np.argmin)
__device__ __forceinline__ int km_best_trial(const KmChain& a, double* pots) {
  if ((int)threadIdx.x < a.trials) pots[threadIdx.x] = km_ordered_sum(a.pT + threadIdx.x, 8, a.G);
  __syncthreads();
  int best = 0;
  double bp = pots[0];
  for (int t = 1; t < a.trials; ++t)
    if (pots[t] < bp) {
      bp = pots[t];
      best = t;
    }
  return best;
}

// offset = sum of src[g * stride] over the workgroups before this one, total = over all of
// them, both added in workgroup order (all loads in flight together, staged through LDS)
__device__ __forceinline__ void km_prefix(const double* src, int stride, int G, double* part,
                                          double* offset, double* total) {
  for (int g = threadIdx.x; g < G; g += kKmT) part[g] = src[(size_t)g * stride];
  __syncthreads();
  if (threadIdx.x == 0) {
    double off = 0.0, tot = 0.0;
    for (int g = 0; g < G; ++g) {
      if (g == (int)blockIdx.x) off = tot;
      tot += part[g];
    }
    part[kKmMaxG] = off;
    part[kKmMaxG + 1] = tot;
  }
  __syncthreads();
  *offset = part[kKmMaxG];
  *total = part[kKmMaxG + 1];
}

// k-means++ round c (1 <= c < k): sample `trials` candidate rows with probability
// proportional to `closest` (sklearn _kmeans_plusplus: searchsorted(stable_cumsum(closest),
// rand * pot)).
template <int KC>
__device__ __forceinline__ void km_select_body(const KmChain& a, int c) {
  __shared__ double mean[KC], crow[KC], sm[kKmW], pots[8], rvals[8], scan[kKmT];
  __shared__ double part[kKmMaxG + 2];
  __shared__ double csq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x * kKmT + tid;
  double cl = r < a.n ? a.closest[r] : 0.0;
  double offset = 0.0, pot = 0.0;
  if (c >= 2) {
    // ---- the previous round's winner becomes centre c - 1
    km_mean(a, mean);
    double v[KC];
    km_load_row<KC>(a, r, v);
    const double xs = r < a.n ? a.xsq[r] : 0.0;
    __syncthreads();
    const int best = km_best_trial(a, pots);
    const int row = a.cand[(c - 1) & 1][best];
    if (tid < a.k) crow[tid] = a.ET[(size_t)tid * a.lde + row] - mean[tid];
    __syncthreads();
    if (tid == 0) {
      double s2 = 0.0;
      for (int j = 0; j < a.k; ++j) s2 += crow[j] * crow[j];
      csq = s2;
      if (blockIdx.x == 0) a.seeds[c - 1] = row;
    }
    __syncthreads();
    if (r < a.n) {
      cl = fmin(cl, km_dist<KC>(v, mean, crow, csq, xs, a.k));
      a.closest[r] = cl;
    }
    // per-workgroup sums of the new `closest` are what km_trials left for this trial
    km_prefix(a.pT + best, 8, a.G, part, &offset, &pot);
  } else {
    km_prefix(a.ppot, 1, a.G, part, &offset, &pot);
  }
  if (tid < a.trials) rvals[tid] = a.rnd[(c - 1) * a.trials + tid] * pot;
  // ---- cumulative sum in row order: scan inside the wave, wave totals, workgroup offset
  double v = cl;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  __syncthreads();
  if (lane == 63) sm[wave] = v;
  __syncthreads();
  double woff = 0.0;
  for (int w = 0; w < wave; ++w) woff += sm[w];
  scan[tid] = v + woff;
  __syncthreads();
  if (r < a.n) {
    const double incl = offset + scan[tid];
    const double excl = tid == 0 ? offset : offset + scan[tid - 1];
    for (int t = 0; t < a.trials; ++t) {
      const double rv = rvals[t];
      // searchsorted(cumsum, rv, 'left'): first index with cumsum >= rv
      if ((rv > excl || r == 0) && rv <= incl) atomicMin(&a.cand[c & 1][t], r);
    }
  }
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_select(const KmChain a, int c) {
  km_select_body<KC>(a, c);
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_select_g(const GroupOf<KmChain> g, int c) {
  const KmChain& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.G || c >= a.k) return;
  km_select_body<KC>(a, c);
}

template <int KC>
__device__ __forceinline__ void km_trials_body(const KmChain& a, int c) {
  __shared__ double mean[KC], crow[8][KC], csq[8], sm[kKmW];
  const int tid = threadIdx.x, r = blockIdx.x * kKmT + tid;
  km_mean(a, mean);
  double v[KC];
  km_load_row<KC>(a, r, v);
  const double xs = r < a.n ? a.xsq[r] : 0.0;
  const double cl = r < a.n ? a.closest[r] : 0.0;
  __syncthreads();
  for (int e = tid; e < a.trials * a.k; e += kKmT) {
    const int t = e / a.k, j = e - t * a.k;
    crow[t][j] = a.ET[(size_t)j * a.lde + a.cand[c & 1][t]] - mean[j];
  }
  __syncthreads();
  if (tid < a.trials) {
    double s2 = 0.0;
    for (int j = 0; j < a.k; ++j) s2 += crow[tid][j] * crow[tid][j];
    csq[tid] = s2;
  }
  __syncthreads();
  for (int t = 0; t < a.trials; ++t) {
    double d = 0.0;
    if (r < a.n) d = fmin(cl, km_dist<KC>(v, mean, crow[t], csq[t], xs, a.k));
    const double tot = km_bsum(d, sm);
    if (tid == 0) a.pT[(size_t)blockIdx.x * 8 + t] = tot;
  }
  if (blockIdx.x == 0 && tid < 8) a.cand[(c + 1) & 1][tid] = a.n - 1;
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_trials(const KmChain a, int c) {
  km_trials_body<KC>(a, c);
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_trials_g(const GroupOf<KmChain> g, int c) {
  const KmChain& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.G || c >= a.k) return;
  km_trials_body<KC>(a, c);
}

// Partial sums of one assignment: for every cluster the member count, the count of members
// with row index > 0 (the reference tests `.any()` on the member INDICES, :137-138) and the
// column sums of the members (x - shift), members added in row order.
template <int KC>
__device__ __forceinline__ void km_cluster_partials(const KmChain& a, const double (&v)[KC],
                                                    const double* shift, int label, int r,
                                                    double* vals, int* labs, double* out) {
  const int tid = threadIdx.x, k = a.k;
  labs[tid] = r < a.n ? label : -1;
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (j < k) vals[tid * KC + j] = v[j] - shift[j];
  __syncthreads();
  // entry e of the (k*k + 2k) sums, split over 4 row segments of 64 rows each; a segment
  // adds its members in row order and the segments are added 0 + 1 + 2 + 3: fixed order
  const int r0 = blockIdx.x * kKmT;
  const int nsum = k * k + 2 * k;
  for (int base = 0; base < nsum; base += kKmT / 4) {
    const int e = base + (tid >> 2), seg = tid & 3;
    double acc = 0.0;
    if (e < nsum) {
      const int t0 = seg * (kKmT / 4);
      if (e < k * k) {
        const int c2 = e / k, j = e - c2 * k;
#pragma unroll 8
        for (int t = t0; t < t0 + kKmT / 4; ++t) acc += labs[t] == c2 ? vals[t * KC + j] : 0.0;
      } else if (e < k * k + k) {
        const int c2 = e - k * k;
#pragma unroll 8
        for (int t = t0; t < t0 + kKmT / 4; ++t) acc += labs[t] == c2 ? 1.0 : 0.0;
      } else {
        const int c2 = e - k * k - k;
#pragma unroll 8
        for (int t = t0; t < t0 + kKmT / 4; ++t)
          acc += (labs[t] == c2 && r0 + t > 0) ? 1.0 : 0.0;
      }
    }
    const double a1 = __shfl_down(acc, 1), a2 = __shfl_down(acc, 2), a3 = __shfl_down(acc, 3);
    if (e < nsum && seg == 0) out[e] = ((acc + a1) + a2) + a3;
  }
}

template <int KC>
__device__ __forceinline__ void km_lloyd_body(const KmChain& a) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* vals = dyn;                                   // kKmT x KC
  int* labs = reinterpret_cast<int*>(vals + kKmT * KC);  // kKmT
  __shared__ double mean[KC], cent[KC * KC], cnorm[KC], pots[8];
  __shared__ int seeds[KC];
  const int tid = threadIdx.x, r = blockIdx.x * kKmT + tid, k = a.k;
  km_mean(a, mean);
  double v[KC];
  km_load_row<KC>(a, r, v);
  __syncthreads();
  if (k > 1) {
    const int best = km_best_trial(a, pots);
    if (tid == 0) {
      seeds[k - 1] = a.cand[(k - 1) & 1][best];
      if (blockIdx.x == 0) a.seeds[k - 1] = seeds[k - 1];
    }
  }
  if (tid < k - 1 || (k == 1 && tid == 0)) seeds[tid] = a.seeds[tid];
  __syncthreads();
  for (int e = tid; e < k * k; e += kKmT) {
    const int c2 = e / k, j = e - c2 * k;
    cent[e] = a.ET[(size_t)j * a.lde + seeds[c2]] - mean[j];
  }
  __syncthreads();
  if (tid < k) {
    double s2 = 0.0;
    for (int j = 0; j < k; ++j) s2 += cent[tid * k + j] * cent[tid * k + j];
    cnorm[tid] = s2;
  }
  __syncthreads();
  int best = 0;
  double bd = INFINITY;
  for (int c2 = 0; c2 < k; ++c2) {
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < KC; ++j)
      if (j < k) dot += (v[j] - mean[j]) * cent[c2 * k + j];
    const double d = cnorm[c2] - 2.0 * dot;
    if (d < bd) {
      bd = d;
      best = c2;
    }
  }
  km_cluster_partials<KC>(a, v, mean, best, r, vals, labs,
                          a.pS[0] + (size_t)blockIdx.x * (k * k + 2 * k));
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_lloyd(const KmChain a) {
  km_lloyd_body<KC>(a);
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_lloyd_g(const GroupOf<KmChain> g) {
  const KmChain& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.G) return;
  km_lloyd_body<KC>(a);
}

// Cosine iteration `it` of CustomKMeans.predict (:118-141); see the header.
template <int KC>
__device__ __forceinline__ void km_cosine_body(const KmChain& a, int it) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* vals = dyn;
  int* labs = reinterpret_cast<int*>(vals + kKmT * KC);
  __shared__ double mean[KC], cent[KC * KC], cnorm[KC], tot[KC * KC + 2 * KC], sm[kKmW];
  __shared__ double zero[KC];
  __shared__ int seeds[KC];
  __shared__ int s_done;
  const int tid = threadIdx.x, r = blockIdx.x * kKmT + tid, k = a.k;
  if (a.info[8] != 0) return;  // an earlier launch met the stop rule
  const int nsum = k * k + 2 * k;
  double v[KC];
  km_load_row<KC>(a, r, v);
  if (tid < KC) zero[tid] = 0.0;
  // ---- the stop rule of iteration it - 1
  if (it > 0) {
    if (tid == 0) {
      const double mean_d = km_ordered_sum(a.pD[it & 1], 1, a.G) / (double)a.n;
      const double prev = it >= 2 ? a.meand[it - 2] : 0.0;
      s_done = ((mean_d <= prev && mean_d >= (1.0 - 0.001) * prev) || it - 1 == a.max_iter)
                   ? 1 : 0;
      if (blockIdx.x == 0) {
        a.meand[it - 1] = mean_d;
        if (s_done) {
          a.info[0] = it;  // iterations run = (it - 1) + 1
          a.info[8] = 1;
        }
      }
    }
    __syncthreads();
    if (s_done) {
      if (blockIdx.x == 0)
        for (int e = tid; e < k * k; e += kKmT)
          a.centroids_out[e] = a.cent[(it - 1) & 1][e];
      return;
    }
  }
  // ---- centroid update from the previous assignment's partial sums
  for (int e = tid; e < nsum; e += kKmT) tot[e] = km_ordered_sum(a.pS[it & 1] + e, nsum, a.G);
  if (it == 0) {
    km_mean(a, mean);
    if (tid < k) seeds[tid] = a.seeds[tid];
    __syncthreads();
    // Lloyd step on the centred data: mean of the members + mean; an empty cluster keeps
    // its seed
    for (int e = tid; e < k * k; e += kKmT) {
      const int c2 = e / k, j = e - c2 * k;
      const double count = tot[k * k + c2];
      const double seedv = a.ET[(size_t)j * a.lde + seeds[c2]] - mean[j];
      cent[e] = (count > 0.0 ? tot[e] / count : seedv) + mean[j];
    }
  } else {
    __syncthreads();
    for (int e = tid; e < k * k; e += kKmT) {
      const int c2 = e / k;
      const double count = tot[k * k + c2], nz = tot[k * k + k + c2];
      cent[e] = nz > 0.0 ? tot[e] / count : a.cent[(it - 1) & 1][e];
    }
  }
  __syncthreads();
  if (blockIdx.x == 0)
    for (int e = tid; e < k * k; e += kKmT) a.cent[it & 1][e] = cent[e];
  if (tid < k) {
    double s2 = 0.0;
    for (int j = 0; j < k; ++j) s2 += cent[tid * k + j] * cent[tid * k + j];
    cnorm[tid] = sqrt(s2);
  }
  __syncthreads();
  // ---- cosine assignment: 1 - clip(e.c / (|e| |c|)); argmin takes the first minimum
  double en = 0.0;
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (j < k) en += v[j] * v[j];
  en = sqrt(en);
  int best = 0;
  double bd = INFINITY;
  for (int c2 = 0; c2 < k; ++c2) {
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < KC; ++j)
      if (j < k) dot += v[j] * cent[c2 * k + j];
    double cosine = dot / (en * cnorm[c2]);
    if (fabs(cosine) > 1.0) cosine = copysign(1.0, cosine);
    const double d = 1.0 - cosine;
    if (d < bd) {
      bd = d;
      best = c2;
    }
  }
  if (r < a.n) a.labels64[r] = best;
  const double dsum = km_bsum(r < a.n ? bd : 0.0, sm);
  if (tid == 0) a.pD[(it + 1) & 1][blockIdx.x] = dsum;
  km_cluster_partials<KC>(a, v, zero, best, r, vals, labs,
                          a.pS[(it + 1) & 1] + (size_t)blockIdx.x * nsum);
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_cosine(const KmChain a, int it) {
  km_cosine_body<KC>(a, it);
}
template <int KC>
__global__ __launch_bounds__(kKmT) void km_cosine_g(const GroupOf<KmChain> g, int it) {
  const KmChain& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.G) return;
  km_cosine_body<KC>(a, it);
}

// ---------------------------------------------------------------- host side
size_t kmeans_chain_workspace_doubles(int n) {
  const size_t G = (size_t)(n + kKmT - 1) / kKmT;
  // pm, ppot, pT, 2 x pS (k <= 32), 2 x pD, meand, 2 x cent, ints (seeds, cand, as doubles)
  return G * 64 + 64 + G + G * 8 + 2 * G * (32 * 32 + 64) + 2 * G + 512 + 2 * 32 * 32 + 256;
}

bool kmeans_chain_supported(int n, int k, int trials) {
  if (k < 1 || k > 32 || trials > 8 || n < 1) return false;
  const size_t G = (size_t)(n + kKmT - 1) / kKmT;
  // every launch adds the G partials of its predecessor in its prologue
  return G <= (size_t)kKmMaxG && G * (size_t)(k * k + 2 * k) <= 65536;
}

static KmChain km_args(const double* ET, int lde, int n, int k, int max_iter, int first_center,
                       int trials, const KmeansWorkspace& ws) {
  KmChain a;
  a.ET = ET;
  a.lde = lde;
  a.n = n;
  a.k = k;
  a.G = (n + kKmT - 1) / kKmT;
  a.rnd = ws.rnd;
  a.trials = trials;
  a.first_center = first_center;
  a.max_iter = max_iter;
  a.closest = ws.closest;
  a.xsq = ws.xsq;
  a.labels64 = ws.labels64;
  a.centroids_out = ws.centroids;
  const size_t G = a.G;
  double* p = ws.chain;
  a.pm = p;          p += G * 64;
  a.meang = p;       p += 64;
  a.ppot = p;        p += G;
  a.pT = p;          p += G * 8;
  a.pS[0] = p;       p += G * (32 * 32 + 64);
  a.pS[1] = p;       p += G * (32 * 32 + 64);
  a.pD[0] = p;       p += G;
  a.pD[1] = p;       p += G;
  a.meand = p;       p += 512;
  a.cent[0] = p;     p += 32 * 32;
  a.cent[1] = p;     p += 32 * 32;
  int* ip = reinterpret_cast<int*>(p);
  a.seeds = ip;
  a.cand[0] = ip + 64;
  a.cand[1] = ip + 72;
  a.info = ws.info;
  return a;
}

template <int KC>
static void km_enqueue(hipStream_t s, const KmChain& a, int it_begin, int it_count) {
  const dim3 grid(a.G), block(kKmT);
  const size_t dyn = sizeof(double) * kKmT * KC + sizeof(int) * kKmT;
  static std::once_flag once[16];
  int dev = 0;
  hipGetDevice(&dev);
  std::call_once(once[dev & 15], [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(km_lloyd<KC>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(km_cosine<KC>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  });
  if (it_begin == 0) {
    hipLaunchKernelGGL(km_colsum<KC>, grid, block, 0, s, a);
    hipLaunchKernelGGL(km_meanreduce, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(km_first<KC>, grid, block, 0, s, a);
    for (int c = 1; c < a.k; ++c) {
      hipLaunchKernelGGL(km_select<KC>, grid, block, 0, s, a, c);
      hipLaunchKernelGGL(km_trials<KC>, grid, block, 0, s, a, c);
    }
    hipLaunchKernelGGL(km_lloyd<KC>, grid, block, dyn, s, a);
  }
  for (int it = it_begin; it < it_begin + it_count; ++it)
    hipLaunchKernelGGL(km_cosine<KC>, grid, block, dyn, s, a, it);
}

template <int KC>
static void km_enqueue_group(hipStream_t s, const GroupOf<KmChain>& g, int count, int gmax,
                             int kmax, int it_begin, int it_count) {
  const dim3 grid(gmax, count), block(kKmT);
  const size_t dyn = sizeof(double) * kKmT * KC + sizeof(int) * kKmT;
  SC_OPT_IN_LDS(km_lloyd_g<KC>, 96 * 1024);
  SC_OPT_IN_LDS(km_cosine_g<KC>, 96 * 1024);
  if (it_begin == 0) {
    hipLaunchKernelGGL(km_colsum_g<KC>, grid, block, 0, s, g);
    hipLaunchKernelGGL(km_meanreduce_g, dim3(1, count), dim3(64), 0, s, g);
    hipLaunchKernelGGL(km_first_g<KC>, grid, block, 0, s, g);
    for (int c = 1; c < kmax; ++c) {  // members with k <= c sit the round out
      hipLaunchKernelGGL(km_select_g<KC>, grid, block, 0, s, g, c);
      hipLaunchKernelGGL(km_trials_g<KC>, grid, block, 0, s, g, c);
    }
    hipLaunchKernelGGL(km_lloyd_g<KC>, grid, block, dyn, s, g);
  }
  for (int it = it_begin; it < it_begin + it_count; ++it)
    hipLaunchKernelGGL(km_cosine_g<KC>, grid, block, dyn, s, g, it);
}

// The same chain for up to kGroupMax independent problems per launch (items[z].n = 0: idle
// member).  Every member runs exactly the arithmetic of launch_kmeans_chain: the register
// tile width KC (taken from the largest k of the group) only sizes arrays, the sums run over
// j < k in the same order.
void launch_kmeans_chain_group(hipStream_t s, const KmGroupItem* items, int count, int it_begin,
                               int it_count) {
  GroupOf<KmChain> g;
  memset(&g, 0, sizeof(g));
  int gmax = 0, kmax = 0;
  for (int z = 0; z < count; ++z) {
    const KmGroupItem& it = items[z];
    if (it.n <= 0) continue;
    g.s[z] = km_args(it.ET, it.lde, it.n, it.k, it.max_iter, it.first_center, it.trials, it.ws);
    gmax = std::max(gmax, g.s[z].G);
    kmax = std::max(kmax, it.k);
  }
  if (gmax == 0) return;
  if (kmax <= 8) km_enqueue_group<8>(s, g, count, gmax, kmax, it_begin, it_count);
  else if (kmax <= 16) km_enqueue_group<16>(s, g, count, gmax, kmax, it_begin, it_count);
  else km_enqueue_group<32>(s, g, count, gmax, kmax, it_begin, it_count);
}

// Enqueue the chain: seeding + Lloyd step (when it_begin == 0) and cosine iterations
// [it_begin, it_begin + it_count).  The caller reads ws.info[8] (done) after a sync and calls
// again with the next iteration range while it is 0.  (Iteration it tests the stop rule of
// it - 1, so max_iter + 2 launches always suffice.)
void launch_kmeans_chain(hipStream_t s, const double* ET, int lde, int n, int k, int max_iter,
                         int first_center, int trials, const KmeansWorkspace& ws, int it_begin,
                         int it_count) {
  const KmChain a = km_args(ET, lde, n, k, max_iter, first_center, trials, ws);
  if (k <= 8) km_enqueue<8>(s, a, it_begin, it_count);
  else if (k <= 16) km_enqueue<16>(s, a, it_begin, it_count);
  else km_enqueue<32>(s, a, it_begin, it_count);
}

}  // namespace sc
