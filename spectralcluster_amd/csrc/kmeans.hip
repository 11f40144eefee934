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

__global__ __launch_bounds__(256) void k_row_renorm(double* __restrict__ E, int lde,
                                                    int n, int k) {
  // spectral_clusterer.py:301-305: rows of the spectral embedding to unit L2 norm
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
  for (int j = 0; j < k; ++j) s += E[(size_t)r * lde + j] * E[(size_t)r * lde + j];
  const double nrm = sqrt(s);
  for (int j = 0; j < k; ++j) E[(size_t)r * lde + j] = E[(size_t)r * lde + j] / nrm;
}

__global__ __launch_bounds__(KT) void k_kmeans(
    const double* __restrict__ E, int lde, int n, int k, int max_iter,
    int first_center, int trials, double* __restrict__ Xc, double* __restrict__ xsq,
    double* __restrict__ closest, double* __restrict__ cand_d,
    double* __restrict__ enorm, const double* __restrict__ rnd,
    double* __restrict__ cent_out, int* __restrict__ labels32,
    long long* __restrict__ labels64, int* __restrict__ info) {
  __shared__ double sm[KW];
  __shared__ double mean[kMaxVectors];
  __shared__ double cent[kMaxVectors * kMaxVectors];   // k x k, stride k
  __shared__ double cnorm[kMaxVectors];
  __shared__ double scan[KT];
  __shared__ double pots[8];
  __shared__ double rvals[8];
  __shared__ int cand[8];
  __shared__ int seeds[kMaxVectors];
  __shared__ int counts[kMaxVectors];
  __shared__ int nzcounts[kMaxVectors];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;

  // ---- column means (numpy mean(axis=0)) and centred copy -------------------
  for (int j = wave; j < k; j += KW) {
    double s = 0.0;
    for (int r = lane; r < n; r += 64) s += E[(size_t)r * lde + j];
    s = wsum(s);
    if (lane == 0) mean[j] = s / (double)n;
  }
  __syncthreads();
  for (int r = tid; r < n; r += KT) {
    double s = 0.0;
    double en = 0.0;
    for (int j = 0; j < k; ++j) {
      const double e = E[(size_t)r * lde + j];
      const double v = e - mean[j];
      Xc[(size_t)r * k + j] = v;
      s += v * v;
      en += e * e;
    }
    xsq[r] = s;
    enorm[r] = sqrt(en);
  }
  __syncthreads();

  // ---- k-means++ (sklearn _kmeans_plusplus, unit sample weights) --------------
  if (tid == 0) seeds[0] = first_center;
  __syncthreads();
  double pot;
  {
    const double* c0 = Xc + (size_t)first_center * k;
    const double csq = xsq[first_center];
    double part = 0.0;
    for (int r = tid; r < n; r += KT) {
      double dot = 0.0;
      for (int j = 0; j < k; ++j) dot += c0[j] * Xc[(size_t)r * k + j];
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
    // distances to the candidates, potentials
    for (int t = 0; t < trials; ++t) {
      const int ci = cand[t];
      const double* cc = Xc + (size_t)ci * k;
      const double csq = xsq[ci];
      double part = 0.0;
      for (int r = tid; r < n; r += KT) {
        double dot = 0.0;
        for (int j = 0; j < k; ++j) dot += cc[j] * Xc[(size_t)r * k + j];
        double d = -2.0 * dot;
        d += csq;
        d += xsq[r];
        d = fmax(d, 0.0);
        d = fmin(closest[r], d);
        cand_d[(size_t)t * n + r] = d;
        part += d;
      }
      const double tot = bsum(part, sm);
      if (tid == 0) pots[t] = tot;
    }
    __syncthreads();
    int best = 0;
    for (int t = 1; t < trials; ++t)
      if (pots[t] < pots[best]) best = t;  // np.argmin: first minimum
    pot = pots[best];
    for (int r = tid; r < n; r += KT) closest[r] = cand_d[(size_t)best * n + r];
    if (tid == 0) seeds[c] = cand[best];
    __syncthreads();
  }

  // ---- one Euclidean Lloyd step on the centred data (max_iter = 1) -------------
  for (int e = tid; e < k * k; e += KT) {
    const int c = e / k, j = e - c * k;
    cent[e] = Xc[(size_t)seeds[c] * k + j];
  }
  __syncthreads();
  if (tid < k) {
    double s = 0.0;
    for (int j = 0; j < k; ++j) s += cent[tid * k + j] * cent[tid * k + j];
    cnorm[tid] = s;  // squared norms here
  }
  __syncthreads();
  for (int r = tid; r < n; r += KT) {
    int best = 0;
    double bd = INFINITY;
    for (int c = 0; c < k; ++c) {
      double dot = 0.0;
      for (int j = 0; j < k; ++j) dot += Xc[(size_t)r * k + j] * cent[c * k + j];
      const double d = cnorm[c] - 2.0 * dot;
      if (d < bd) { bd = d; best = c; }
    }
    labels32[r] = best;
  }
  __syncthreads();
  for (int c = wave; c < k; c += KW) {
    int cnt = 0;
    for (int r = lane; r < n; r += 64) cnt += labels32[r] == c;
    cnt = wsumi(cnt);
    if (lane == 0) counts[c] = cnt;
  }
  __syncthreads();
  for (int q = wave; q < k * k; q += KW) {
    const int c = q / k, j = q - c * k;
    double s = 0.0;
    for (int r = lane; r < n; r += 64)
      if (labels32[r] == c) s += Xc[(size_t)r * k + j];
    s = wsum(s);
    if (lane == 0) {
      // empty cluster: keep the seed (sklearn relocates; unreachable from
      // k-means++ seeds, each of which is its own nearest centre)
      const double v = counts[c] > 0 ? s / (double)counts[c] : cent[q];
      cent[q] = v + mean[j];  // best_centers += X_mean
    }
  }
  __syncthreads();

  // ---- CustomKMeans.predict, cosine (custom_distance_kmeans.py:118-141) --------
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
    for (int r = tid; r < n; r += KT) {
      int best = 0;
      double bd = INFINITY;
      const double nu = enorm[r];
      for (int c = 0; c < k; ++c) {
        double dot = 0.0;
        for (int j = 0; j < k; ++j) dot += E[(size_t)r * lde + j] * cent[c * k + j];
        double cosine = dot / (nu * cnorm[c]);
        if (fabs(cosine) > 1.0) cosine = copysign(1.0, cosine);
        const double d = 1.0 - cosine;
        if (d < bd) { bd = d; best = c; }
      }
      labels32[r] = best;
      part += bd;
    }
    const double mean_d = bsum(part, sm) / (double)n;
    // (:131-133)
    if ((mean_d <= prev && mean_d >= (1.0 - 0.001) * prev) || it == max_iter) break;
    prev = mean_d;
    for (int c = wave; c < k; c += KW) {
      int cnt = 0, nz = 0;
      for (int r = lane; r < n; r += 64) {
        const int hit = labels32[r] == c;
        cnt += hit;
        nz += hit && r > 0;
      }
      cnt = wsumi(cnt);
      nz = wsumi(nz);
      if (lane == 0) { counts[c] = cnt; nzcounts[c] = nz; }
    }
    __syncthreads();
    for (int q = wave; q < k * k; q += KW) {
      const int c = q / k, j = q - c * k;
      if (nzcounts[c] == 0) continue;  // `.any()` on the member INDICES (:137-138)
      double s = 0.0;
      for (int r = lane; r < n; r += 64)
        if (labels32[r] == c) s += E[(size_t)r * lde + j];
      s = wsum(s);
      if (lane == 0) cent[q] = s / (double)counts[c];
    }
    __syncthreads();
  }
  for (int r = tid; r < n; r += KT) labels64[r] = labels32[r];
  for (int e = tid; e < k * k; e += KT) cent_out[e] = cent[e];
  if (tid == 0) info[0] = it + 1;
}

void launch_row_renorm(hipStream_t s, double* E, int lde, int n, int k) {
  hipLaunchKernelGGL(k_row_renorm, dim3((n + 255) / 256), dim3(256), 0, s, E, lde, n,
                     k);
}

void launch_kmeans(hipStream_t s, const double* E, int lde, int n, int k,
                   int max_iter, int first_center, int trials,
                   const KmeansWorkspace& ws) {
  hipLaunchKernelGGL(k_kmeans, dim3(1), dim3(KT), 0, s, E, lde, n, k, max_iter,
                     first_center, trials, ws.Xc, ws.xsq, ws.closest, ws.cand,
                     ws.enorm, ws.rnd, ws.centroids, ws.labels32, ws.labels64,
                     ws.info);
}

}  // namespace sc
