// K1/K2: run_kmeans with custom_dist="cosine" (reference
// custom_distance_kmeans.py:13-141) on the (n, k) spectral embedding, k <= 64.
//
//   seeds   = sklearn 1.7.2 KMeans(init="k-means++", max_iter=1, random_state=0,
//             n_init="auto").fit(E).cluster_centers_   (:39-43), restated:
//             centre E, k-means++ with RandomState(0) doubles (host MT19937,
//             passed in `rnd`), one Euclidean Lloyd step, add the mean back;
//   loop    = CustomKMeans.predict (:85-141): cosine cdist, argmin, mean
//             distance stop rule, centroid means incl. the `.any()`-on-indices
//             quirk (:137-138).
//
// The whole stage is ONE single-workgroup kernel (1024 threads): the data is at
// most n*k*8 = 1.3 MB at n = 8192, k = 20 and L2-resident; a grid would spend its
// time in launch gaps, not arithmetic.  Compiled with -ffp-contract=off.
#include "sc_internal.h"

namespace sc {

constexpr int KT = 1024;  // threads
constexpr int KW = KT / 64;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wsumi(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double bsum(double v, double* sm) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < KW; ++w) t += sm[w];
  return t;
}

// E is stored COLUMN-MAJOR on the device (one eigenvector = one contiguous run of
// n doubles, ET[j * lde + r]): every per-row loop below then reads 512 contiguous
// bytes per wave instruction instead of 64 scattered lines.
__global__ __launch_bounds__(256) void k_row_renorm(double* __restrict__ ET, int lde,
                                                    int n, int k) {
  // spectral_clusterer.py:301-305: rows of the spectral embedding to unit L2 norm
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
  for (int j = 0; j < k; ++j) s += ET[(size_t)j * lde + r] * ET[(size_t)j * lde + r];
  const double nrm = sqrt(s);
  for (int j = 0; j < k; ++j) ET[(size_t)j * lde + r] = ET[(size_t)j * lde + r] / nrm;
}

// (n, k) row-major  ->  column-major with leading dimension ldt (stage API input)
__global__ void k_to_colmajor(const double* __restrict__ src, int n, int k,
                              double* __restrict__ dst, int ldt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  const int r = e / k, j = e - r * k;
  dst[(size_t)j * ldt + r] = src[e];
}

// Per-cluster means of the member rows, data column-major (data[j * ld + r]).
// Each thread owns rows tid, tid + KT, ...; for one cluster it adds up only its own
// member rows, then the k partial sums are reduced wave -> LDS -> fixed-order total
// (deterministic).
//   mode 0 (Lloyd, centred data):  mean of members + mean[j]; empty cluster keeps seed
//   mode 1 (cosine loop):          mean of members iff some member index > 0
__device__ __forceinline__ void cluster_means(const double* __restrict__ data, int ld,
                                              int n, int k, const int* __restrict__ lab,
                                              double* cent, const double* mean, int mode,
                                              double* wpart /* 2 * KW * kMaxVectors */,
                                              int* counts, int* nzcounts) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = 0; c < k; ++c) {
    int cnt = 0, nz = 0;
    double* wp = wpart + (c & 1) * (KW * kMaxVectors);  // double-buffered by parity
    for (int j0 = 0; j0 < k; j0 += 8) {
      double acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = 0.0;
#pragma unroll 4
      for (int r = tid; r < n; r += KT) {
        if (lab[r] == c) {
          if (j0 == 0) { ++cnt; nz += r > 0; }
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (j0 + q < k) acc[q] += data[(size_t)(j0 + q) * ld + r];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double v = wsum(acc[q]);
        if (lane == 0 && j0 + q < k) wp[wave * kMaxVectors + j0 + q] = v;
      }
    }
    cnt = wsumi(cnt);
    nz = wsumi(nz);
    if (lane == 0) {
      atomicAdd(&counts[c], cnt);   // integer: order-independent
      atomicAdd(&nzcounts[c], nz);
    }
    __syncthreads();
    if (tid < k) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < KW; ++w) tot += wp[w * kMaxVectors + tid];
      const int q = c * k + tid;
      if (mode == 0) {
        const double v = counts[c] > 0 ? tot / (double)counts[c] : cent[q];
        cent[q] = v + mean[tid];
      } else if (nzcounts[c] > 0) {
        cent[q] = tot / (double)counts[c];
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(KT) void k_kmeans(
    const double* __restrict__ ET, int lde, int n, int k, int max_iter,
    int first_center, int trials, double* __restrict__ XcT, double* __restrict__ xsq,
    double* __restrict__ closest, double* __restrict__ cand_d,
    double* __restrict__ enorm, const double* __restrict__ rnd,
    double* __restrict__ cent_out, int* __restrict__ labels32,
    long long* __restrict__ labels64, int* __restrict__ info) {
  __shared__ double sm[KW];
  __shared__ double mean[kMaxVectors];
  __shared__ double cent[kMaxVectors * kMaxVectors];   // k x k, stride k
  __shared__ double cnorm[kMaxVectors];
  __shared__ double candrow[8 * kMaxVectors];          // candidate rows, stride k
  __shared__ double candsq[8];
  __shared__ double scan[KT];
  __shared__ double pots[8];
  __shared__ double rvals[8];
  __shared__ int cand[8];
  __shared__ int seeds[kMaxVectors];
  __shared__ int counts[kMaxVectors];
  __shared__ int nzcounts[kMaxVectors];
  __shared__ double wpart[2 * KW * kMaxVectors];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const long long t_start = wall_clock64();  // 100 MHz; phase times go to info[1..4]

  // ---- column means (numpy mean(axis=0)), centred copy, row norms ---------------
  for (int j = wave; j < k; j += KW) {
    double s = 0.0;
    for (int r = lane; r < n; r += 64) s += ET[(size_t)j * lde + r];
    s = wsum(s);
    if (lane == 0) mean[j] = s / (double)n;
  }
  __syncthreads();
  for (int r = tid; r < n; r += KT) {
    double s = 0.0, en = 0.0;
    for (int j = 0; j < k; ++j) {
      const double e = ET[(size_t)j * lde + r];
      const double v = e - mean[j];
      XcT[(size_t)j * n + r] = v;
      s += v * v;
      en += e * e;
    }
    xsq[r] = s;
    enorm[r] = sqrt(en);
  }
  __syncthreads();

  // ---- k-means++ (sklearn _kmeans_plusplus, unit sample weights) --------------
  if (tid == 0) info[1] = (int)(wall_clock64() - t_start);
  if (tid == 0) seeds[0] = first_center;
  if (tid < k) candrow[tid] = XcT[(size_t)tid * n + first_center];
  __syncthreads();
  double pot;
  {
    const double csq = xsq[first_center];
    double part = 0.0;
    for (int r = tid; r < n; r += KT) {
      double dot = 0.0;
      for (int j = 0; j < k; ++j) dot += candrow[j] * XcT[(size_t)j * n + r];
      double d = -2.0 * dot;
      d += csq;
      d += xsq[r];
      d = fmax(d, 0.0);
      closest[r] = d;
      part += d;
    }
    pot = bsum(part, sm);
  }
  int rpos = 0;
  const int chunk = (n + KT - 1) / KT;
  for (int c = 1; c < k; ++c) {
    if (tid < trials) {
      rvals[tid] = rnd[rpos + tid] * pot;
      cand[tid] = n - 1;  // np.clip(candidate_ids, None, n - 1)
    }
    rpos += trials;
    // inclusive scan of per-thread chunk sums of `closest`
    const int beg = min(n, tid * chunk), end = min(n, beg + chunk);
    double mysum = 0.0;
    for (int r = beg; r < end; ++r) mysum += closest[r];
    {
      double v = mysum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o);
        if (lane >= o) v += u;
      }
      __syncthreads();
      if (lane == 63) sm[wave] = v;
      __syncthreads();
      double off = 0.0;
      for (int w = 0; w < wave; ++w) off += sm[w];
      scan[tid] = v + off;  // inclusive prefix over threads
    }
    __syncthreads();
    {
      const double excl = tid == 0 ? 0.0 : scan[tid - 1];
      const double incl = scan[tid];
      for (int t = 0; t < trials; ++t) {
        const double rv = rvals[t];
        // searchsorted(cumsum, rv, 'left'): first index with cumsum >= rv
        if (beg < end && (rv > excl || tid == 0) && rv <= incl) {
          double run = excl;
          int hit = end - 1;
          for (int r = beg; r < end - 1; ++r) {
            run += closest[r];
            if (run >= rv) { hit = r; break; }
          }
          atomicMin(&cand[t], hit);
        }
      }
    }
    __syncthreads();
    // candidate rows -> LDS, then ONE pass over the data for all trials
    for (int e = tid; e < trials * k; e += KT) {
      const int t = e / k, j = e - t * k;
      candrow[t * k + j] = XcT[(size_t)j * n + cand[t]];
    }
    if (tid < trials) candsq[tid] = xsq[cand[tid]];
    __syncthreads();
    double part[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) part[t] = 0.0;
#pragma unroll 4
    for (int r = tid; r < n; r += KT) {
      double dot[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) dot[t] = 0.0;
      for (int j = 0; j < k; ++j) {
        const double x = XcT[(size_t)j * n + r];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t < trials) dot[t] += candrow[t * k + j] * x;
      }
      const double xs = xsq[r], cl = closest[r];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t < trials) {
          double d = -2.0 * dot[t];
          d += candsq[t];
          d += xs;
          d = fmax(d, 0.0);
          d = fmin(cl, d);
          cand_d[(size_t)t * n + r] = d;
          part[t] += d;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < trials) {  // uniform
        const double tot = bsum(part[t], sm);
        if (tid == 0) pots[t] = tot;
      }
    }
    __syncthreads();
    int best = 0;
    for (int t = 1; t < trials; ++t)
      if (pots[t] < pots[best]) best = t;  // np.argmin: first minimum
    pot = pots[best];
#pragma unroll 4
    for (int r = tid; r < n; r += KT) closest[r] = cand_d[(size_t)best * n + r];
    if (tid == 0) seeds[c] = cand[best];
    __syncthreads();
  }

  // ---- one Euclidean Lloyd step on the centred data (max_iter = 1) -------------
  if (tid == 0) info[2] = (int)(wall_clock64() - t_start);
  for (int e = tid; e < k * k; e += KT) {
    const int c = e / k, j = e - c * k;
    cent[e] = XcT[(size_t)j * n + seeds[c]];
  }
  __syncthreads();
  if (tid < k) {
    double s = 0.0;
    for (int j = 0; j < k; ++j) s += cent[tid * k + j] * cent[tid * k + j];
    cnorm[tid] = s;  // squared norms here
  }
  __syncthreads();
#pragma unroll 2
  for (int r = tid; r < n; r += KT) {
    int best = 0;
    double bd = INFINITY;
    for (int c0 = 0; c0 < k; c0 += 8) {
      double dot[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) dot[q] = 0.0;
      for (int j = 0; j < k; ++j) {
        const double x = XcT[(size_t)j * n + r];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (c0 + q < k) dot[q] += x * cent[(c0 + q) * k + j];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (c0 + q < k) {
          const double d = cnorm[c0 + q] - 2.0 * dot[q];
          if (d < bd) { bd = d; best = c0 + q; }
        }
      }
    }
    labels32[r] = best;
  }
  if (tid < k) { counts[tid] = 0; nzcounts[tid] = 0; }
  __syncthreads();
  // empty cluster: keeps its seed (sklearn relocates; unreachable from k-means++
  // seeds, each of which is its own nearest centre); best_centers += X_mean
  cluster_means(XcT, n, n, k, labels32, cent, mean, 0, wpart, counts, nzcounts);

  // ---- CustomKMeans.predict, cosine (custom_distance_kmeans.py:118-141) --------
  if (tid == 0) info[3] = (int)(wall_clock64() - t_start);
  double prev = 0.0;
  int it = 0;
  for (;; ++it) {
    if (tid < k) {
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += cent[tid * k + j] * cent[tid * k + j];
      cnorm[tid] = sqrt(s);
    }
    __syncthreads();
    double part = 0.0;
#pragma unroll 2
    for (int r = tid; r < n; r += KT) {
      int best = 0;
      double bd = INFINITY;
      const double nu = enorm[r];
      for (int c0 = 0; c0 < k; c0 += 8) {
        double dot[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) dot[q] = 0.0;
        for (int j = 0; j < k; ++j) {
          const double x = ET[(size_t)j * lde + r];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (c0 + q < k) dot[q] += x * cent[(c0 + q) * k + j];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (c0 + q < k) {
            double cosine = dot[q] / (nu * cnorm[c0 + q]);
            if (fabs(cosine) > 1.0) cosine = copysign(1.0, cosine);
            const double d = 1.0 - cosine;
            if (d < bd) { bd = d; best = c0 + q; }
          }
        }
      }
      labels32[r] = best;
      part += bd;
    }
    const double mean_d = bsum(part, sm) / (double)n;
    // (:131-133)
    if ((mean_d <= prev && mean_d >= (1.0 - 0.001) * prev) || it == max_iter) break;
    prev = mean_d;
    if (tid < k) { counts[tid] = 0; nzcounts[tid] = 0; }
    __syncthreads();
    // centroid <- mean of members iff `.any()` of the member INDICES (:137-138)
    cluster_means(ET, lde, n, k, labels32, cent, mean, 1, wpart, counts, nzcounts);
  }
  for (int r = tid; r < n; r += KT) labels64[r] = labels32[r];
  for (int e = tid; e < k * k; e += KT) cent_out[e] = cent[e];
  if (tid == 0) {
    info[0] = it + 1;
    info[4] = (int)(wall_clock64() - t_start);
  }
}

void launch_row_renorm(hipStream_t s, double* ET, int lde, int n, int k) {
  hipLaunchKernelGGL(k_row_renorm, dim3((n + 255) / 256), dim3(256), 0, s, ET, lde, n,
                     k);
}

void launch_to_colmajor(hipStream_t s, const double* src, int n, int k, double* dst,
                        int ldt) {
  hipLaunchKernelGGL(k_to_colmajor, dim3((n * k + 255) / 256), dim3(256), 0, s, src, n, k,
                     dst, ldt);
}

void launch_kmeans(hipStream_t s, const double* ET, int lde, int n, int k,
                   int max_iter, int first_center, int trials,
                   const KmeansWorkspace& ws) {
  hipLaunchKernelGGL(k_kmeans, dim3(1), dim3(KT), 0, s, ET, lde, n, k, max_iter,
                     first_center, trials, ws.Xc, ws.xsq, ws.closest, ws.cand,
                     ws.enorm, ws.rnd, ws.centroids, ws.labels32, ws.labels64,
                     ws.info);
}

}  // namespace sc
