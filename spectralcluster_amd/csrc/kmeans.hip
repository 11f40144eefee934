// K1/K2: run_kmeans with custom_dist="cosine" (reference
// custom_distance_kmeans.py:13-141) on the (n, k) spectral embedding; any k (k <= 64 with the
// per-cluster arrays in LDS, more through a global workspace: k_kmeans<true>).
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
#include <cstdlib>

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
//   mode 1 (custom-distance loop): mean of members iff some member index > 0
__device__ __forceinline__ void cluster_means(const double* __restrict__ data, int ld,
                                              int n, int k, const int* __restrict__ lab,
                                              double* cent, const double* mean, int mode,
                                              double* wpart /* 2 * KW * kst */, int kst,
                                              int* counts, int* nzcounts) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = 0; c < k; ++c) {
    int cnt = 0, nz = 0;
    double* wp = wpart + (c & 1) * (KW * kst);  // double-buffered by parity
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
        if (lane == 0 && j0 + q < k) wp[wave * kst + j0 + q] = v;
      }
    }
    cnt = wsumi(cnt);
    nz = wsumi(nz);
    if (lane == 0) {
      atomicAdd(&counts[c], cnt);   // integer: order-independent
      atomicAdd(&nzcounts[c], nz);
    }
    __syncthreads();
    // (the counters were built by atomics, which execute in L2: in the large-k form they live
    //  in global memory, and a plain load could be served from this CU's L1 with the zero that
    //  was stored before the atomics -- read them the way they were written)
    const int count_c = __hip_atomic_load(&counts[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nz_c = __hip_atomic_load(&nzcounts[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int t = tid; t < k; t += KT) {  // (k <= 64: one trip; the large-k form strides)
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < KW; ++w) tot += wp[w * kst + t];
      const int q = c * k + t;
      if (mode == 0) {
        const double v = count_c > 0 ? tot / (double)count_c : cent[q];
        cent[q] = v + mean[t];
      } else if (nz_c > 0) {
        cent[q] = tot / (double)count_c;
      }
    }
  }
  __syncthreads();
}

// BIG: more than kMaxVectors clusters (the reference has no limit: custom_distance_kmeans.py:
// 13-52 takes any k) -- the per-cluster arrays live in a global workspace `gws` / `gwi` instead
// of LDS (one workgroup: its own stores are visible to it behind __syncthreads), everything else
// is the same code.  gws: k (2 + 8 + 2 KW) + k^2 doubles, gwi: 3 k ints.
// (16 k-means++ trial slots in the BIG form: sklearn draws 2 + int(log k) candidates per centre,
//  more than 8 from k = 1097 on; 16 cover every k a 32-bit sample count allows)
size_t kmeans_big_workspace_doubles(int k) { return (size_t)k * (2 + 16 + 2 * KW) + (size_t)k * k; }
template <bool BIG>
__global__ __launch_bounds__(KT) void k_kmeans(
    const double* __restrict__ ET, int lde, int n, int k, int max_iter,
    int first_center, int trials, double* __restrict__ XcT, double* __restrict__ xsq,
    double* __restrict__ closest, double* __restrict__ cand_d,
    double* __restrict__ enorm, const double* __restrict__ rnd,
    double* __restrict__ cent_out, int* __restrict__ labels32,
    long long* __restrict__ labels64, int* __restrict__ info, int metric,
    double* __restrict__ gws, int* __restrict__ gwi) {
  constexpr int KL = BIG ? 1 : kMaxVectors;  // LDS footprint of the per-cluster arrays
  __shared__ double sm[KW];
  __shared__ double s_mean[KL];
  __shared__ double s_cent[KL * KL];   // k x k, stride k
  __shared__ double s_cnorm[KL];
  constexpr int TS = BIG ? 16 : 8;              // k-means++ trial slots (2 + int(log k) used)
  __shared__ double s_candrow[8 * KL];          // candidate rows, stride k
  __shared__ double candsq[TS];
  __shared__ double scan[KT];
  __shared__ double pots[TS];
  __shared__ double rvals[TS];
  __shared__ int cand[TS];
  __shared__ int s_seeds[KL];
  __shared__ int s_counts[KL];
  __shared__ int s_nzcounts[KL];
  __shared__ double s_wpart[2 * KW * KL];
  const int kst = BIG ? k : kMaxVectors;  // stride of the per-wave partial sums
  double* mean = BIG ? gws : s_mean;
  double* cnorm = BIG ? gws + k : s_cnorm;
  double* candrow = BIG ? gws + 2 * (size_t)k : s_candrow;
  double* wpart = BIG ? gws + (2 + TS) * (size_t)k : s_wpart;
  double* cent = BIG ? gws + (size_t)k * (2 + TS + 2 * KW) : s_cent;
  int* seeds = BIG ? gwi : s_seeds;
  int* counts = BIG ? gwi + k : s_counts;
  int* nzcounts = BIG ? gwi + 2 * (size_t)k : s_nzcounts;
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
  for (int t = tid; t < k; t += KT) candrow[t] = XcT[(size_t)t * n + first_center];
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
    double part[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) part[t] = 0.0;
#pragma unroll 4
    for (int r = tid; r < n; r += KT) {
      double dot[TS];
#pragma unroll
      for (int t = 0; t < TS; ++t) dot[t] = 0.0;
      for (int j = 0; j < k; ++j) {
        const double x = XcT[(size_t)j * n + r];
#pragma unroll
        for (int t = 0; t < TS; ++t)
          if (t < trials) dot[t] += candrow[t * k + j] * x;
      }
      const double xs = xsq[r], cl = closest[r];
#pragma unroll
      for (int t = 0; t < TS; ++t) {
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
    for (int t = 0; t < TS; ++t) {
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
  for (int t = tid; t < k; t += KT) {
    double s = 0.0;
    for (int j = 0; j < k; ++j) s += cent[t * k + j] * cent[t * k + j];
    cnorm[t] = s;  // squared norms here
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
  for (int t = tid; t < k; t += KT) { counts[t] = 0; nzcounts[t] = 0; }
  __syncthreads();
  // empty cluster: keeps its seed (sklearn relocates; unreachable from k-means++
  // seeds, each of which is its own nearest centre); best_centers += X_mean
  cluster_means(XcT, n, n, k, labels32, cent, mean, 0, wpart, kst, counts, nzcounts);

  // ---- CustomKMeans.predict (custom_distance_kmeans.py:118-141), scipy cdist metric --
  if (tid == 0) info[3] = (int)(wall_clock64() - t_start);
  double prev = 0.0;
  int it = 0;
  // scipy's `correlation` is its cosine distance on row-centred operands (scipy 1.15
  // spatial/distance.py _correlation_cdist_wrap: X - X.mean(axis=1, keepdims=True)): the row mean
  // in numpy's summation order for short rows, then the cosine code on the differences
  // (numpy's pairwise_sum, numpy/core/src/umath/loops_utils.h: fewer than 8 elements one by one;
  //  up to PW_BLOCKSIZE = 128 eight running sums folded as a tree, then the tail; longer rows --
  //  more than 128 clusters -- split at n / 2 rounded down to a multiple of 8 and the halves'
  //  sums added, recursively: an explicit stack here, depth <= log2(k / 128) + 1)
  auto leaf_sum = [&](auto at, int lo, int cnt) -> double {
    double s;
    if (cnt < 8) {
      s = 0.0;
      for (int j = 0; j < cnt; ++j) s += at(lo + j);
    } else {
      double a8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a8[q] = at(lo + q);
      int j = 8;
      for (; j < cnt - (cnt % 8); j += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] += at(lo + j + q);
      }
      s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
      for (; j < cnt; ++j) s += at(lo + j);
    }
    return s;
  };
  auto row_mean = [&](auto at) -> double {
    if (k <= 128) return leaf_sum(at, 0, k) / (double)k;
    int lo[24], cnt[24], stage[24];
    double left[24];
    int sp = 0;
    lo[0] = 0;
    cnt[0] = k;
    stage[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
      if (stage[sp] == 0) {
        if (cnt[sp] <= 128) {
          ret = leaf_sum(at, lo[sp], cnt[sp]);
          --sp;
          continue;
        }
        int half = cnt[sp] / 2;
        half -= half % 8;
        stage[sp] = 1;
        lo[sp + 1] = lo[sp];
        cnt[sp + 1] = half;
        stage[sp + 1] = 0;
        ++sp;
      } else if (stage[sp] == 1) {
        left[sp] = ret;
        int half = cnt[sp] / 2;
        half -= half % 8;
        stage[sp] = 2;
        lo[sp + 1] = lo[sp] + half;
        cnt[sp + 1] = cnt[sp] - half;
        stage[sp + 1] = 0;
        ++sp;
      } else {
        ret = left[sp] + ret;
        --sp;
      }
    }
    return ret / (double)k;
  };
  for (;; ++it) {
    for (int t = tid; t < k; t += KT) {
      double s = 0.0;
      if (metric == kKmeansCorrelation) {
        const double mc = row_mean([&](int j) { return cent[t * k + j]; });
        for (int j = 0; j < k; ++j) s += (cent[t * k + j] - mc) * (cent[t * k + j] - mc);
        // (the centred centroid's mean rides in the wave-partial scratch: free at this point)
        wpart[t] = mc;
      } else {
        for (int j = 0; j < k; ++j) s += cent[t * k + j] * cent[t * k + j];
      }
      cnorm[t] = sqrt(s);
    }
    __syncthreads();
    double part = 0.0;
#pragma unroll 2
    for (int r = tid; r < n; r += KT) {
      int best = 0;
      double bd = INFINITY;
      double nu = enorm[r], mx = 0.0;
      if (metric == kKmeansCorrelation) {
        mx = row_mean([&](int j) { return ET[(size_t)j * lde + r]; });
        double s = 0.0;
        for (int j = 0; j < k; ++j) {
          const double e = ET[(size_t)j * lde + r] - mx;
          s += e * e;
        }
        nu = sqrt(s);
      }
      for (int c0 = 0; c0 < k; c0 += 8) {
        double dot[8], aux[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) dot[q] = aux[q] = 0.0;
        for (int j = 0; j < k; ++j) {
          const double x = ET[(size_t)j * lde + r];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (c0 + q < k) {
              const double cv = cent[(c0 + q) * k + j];
              if (metric == kKmeansCosine) {
                dot[q] += x * cv;
              } else if (metric == kKmeansCorrelation) {
                dot[q] += (x - mx) * (cv - wpart[c0 + q]);
              } else if (metric == kKmeansCityblock) {
                dot[q] += fabs(x - cv);
              } else if (metric == kKmeansChebyshev) {
                dot[q] = fmax(dot[q], fabs(x - cv));
              } else if (metric == kKmeansBraycurtis) {  // sum |u - v| / sum |u + v|
                dot[q] += fabs(x - cv);
                aux[q] += fabs(x + cv);
              } else if (metric == kKmeansCanberra) {  // sum |u - v| / (|u| + |v|), 0 / 0 = 0
                const double den = fabs(x) + fabs(cv);
                if (den > 0.0) dot[q] += fabs(x - cv) / den;
              } else {  // (squared) Euclidean
                dot[q] += (x - cv) * (x - cv);
              }
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (c0 + q < k) {
            double d;
            if (metric == kKmeansCosine || metric == kKmeansCorrelation) {
              double cosine = dot[q] / (nu * cnorm[c0 + q]);
              if (fabs(cosine) > 1.0) cosine = copysign(1.0, cosine);
              d = 1.0 - cosine;
            } else if (metric == kKmeansBraycurtis) {
              d = dot[q] / aux[q];
            } else {
              d = metric == kKmeansEuclidean ? sqrt(dot[q]) : dot[q];
            }
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
    for (int t = tid; t < k; t += KT) { counts[t] = 0; nzcounts[t] = 0; }
    __syncthreads();
    // centroid <- mean of members iff `.any()` of the member INDICES (:137-138)
    cluster_means(ET, lde, n, k, labels32, cent, mean, 1, wpart, kst, counts, nzcounts);
  }
  for (int r = tid; r < n; r += KT) labels64[r] = labels32[r];
  for (int e = tid; e < k * k; e += KT) cent_out[e] = cent[e];
  if (tid == 0) {
    info[0] = it + 1;
    info[4] = (int)(wall_clock64() - t_start);
  }
}

// ------------------------------------------------------------------------------------
// Register-resident variant for n <= 1024 * NR (NR = 8 or 16) and k <= 8.
// Thread t owns the NR CONTIGUOUS rows NR*t .. NR*t+NR-1, so
//   * every column access is one 64-byte (NR = 8) vector load per lane, a wave reading
//     4 KiB contiguous: each pass over the data streams from L2 at full line efficiency;
//   * per-row state (closest distance, |x|^2, |e|, label) lives in registers for the
//     whole kernel -- no global round trips between the phases;
//   * the cumulative sum k-means++ samples from is a per-thread running sum in row
//     order plus one block scan of the thread totals;
//   * all k-means++ trials of a round, and all cluster sums of an update, share one pass.
// Same arithmetic as k_kmeans (and as the reference); only the association of the long
// sums differs.
// ------------------------------------------------------------------------------------
template <int NR>
__device__ __forceinline__ void load_rows(const double* __restrict__ col, int r0, int n,
                                          bool full, double (&v)[NR]) {
  if (full) {
#pragma unroll
    for (int q = 0; q < NR / 2; ++q) {
      const double2 d = *reinterpret_cast<const double2*>(col + r0 + 2 * q);
      v[2 * q] = d.x;
      v[2 * q + 1] = d.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = (r0 + i < n) ? col[r0 + i] : 0.0;
  }
}

template <int NR, int NT>
__global__ __launch_bounds__(NT) void k_kmeans_fast(
    const double* __restrict__ ET, int lde, int n, int k, int max_iter,
    int first_center, int trials, const double* __restrict__ rnd,
    double* __restrict__ cent_out, long long* __restrict__ labels64,
    int* __restrict__ info) {
  constexpr int KC = 8;  // k <= 8 on this path
  constexpr int NW = NT / 64;  // waves
  __shared__ double sm[NW * 8];
  __shared__ double mean[KC];
  __shared__ double cent[KC * KC];      // k x k, stride k
  __shared__ double cnorm[KC];
  __shared__ double candrow[8 * KC];    // candidate rows, stride k
  __shared__ double candsq[8];
  __shared__ double scan[NT];
  __shared__ double pots[8];
  __shared__ double rvals[8];
  __shared__ int cand[8];
  __shared__ int seeds[KC];
  __shared__ double wsumv[NW * KC * KC];  // per-wave partial cluster sums
  __shared__ int wcnt[NW * KC], wnz[NW * KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long t_start = wall_clock64();
  const int r0 = NR * tid;
  const bool full = r0 + NR <= n;
  int nvalid = n - r0;
  nvalid = nvalid < 0 ? 0 : (nvalid > NR ? NR : nvalid);

  // block reduction of `cnt` doubles held by every thread (cnt <= 8): result in sm[0..cnt)
  auto block_sum = [&](double* vals, int cnt) {
    for (int q = 0; q < cnt; ++q) {
      const double v = wsum(vals[q]);
      if (lane == 0) sm[wave * 8 + q] = v;
    }
    __syncthreads();
    if (tid < cnt) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += sm[w * 8 + tid];
      pots[tid] = t;  // reuse `pots` as the broadcast slot
    }
    __syncthreads();
  };

  // ---- column means ---------------------------------------------------------------
  {
    double part[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      part[j] = 0.0;
      if (j < k) {
        double v[NR];
        load_rows<NR>(ET + (size_t)j * lde, r0, n, full, v);
#pragma unroll
        for (int i = 0; i < NR; ++i) part[j] += v[i];
      }
    }
    block_sum(part, k);
    if (tid < k) mean[tid] = pots[tid] / (double)n;
    __syncthreads();
  }
  // ---- |x - mean|^2 and |e| per row -------------------------------------------------
  double xsq[NR], closest[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) xsq[i] = 0.0;
  for (int j = 0; j < k; ++j) {
    double v[NR];
    load_rows<NR>(ET + (size_t)j * lde, r0, n, full, v);
    const double mj = mean[j];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const double x = v[i] - mj;
      xsq[i] += x * x;
    }
  }
  if (tid == 0) info[1] = (int)(wall_clock64() - t_start);

  // ---- k-means++ ----------------------------------------------------------------------
  if (tid == 0) seeds[0] = first_center;
  if (tid < k) candrow[tid] = ET[(size_t)tid * lde + first_center] - mean[tid];
  if (tid == 0) {
    // |x_first - mean|^2 in the same association as the owner's xsq
    double s2 = 0.0;
    for (int j = 0; j < k; ++j) {
      const double x = ET[(size_t)j * lde + first_center] - mean[j];
      s2 += x * x;
    }
    candsq[0] = s2;
  }
  __syncthreads();
  double pot;
  {
    double dot[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) dot[i] = 0.0;
    for (int j = 0; j < k; ++j) {
      double v[NR];
      load_rows<NR>(ET + (size_t)j * lde, r0, n, full, v);
      const double mj = mean[j], cj = candrow[j];
#pragma unroll
      for (int i = 0; i < NR; ++i) dot[i] += cj * (v[i] - mj);
    }
    double part = 0.0;
    const double csq = candsq[0];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      double d = -2.0 * dot[i];
      d += csq;
      d += xsq[i];
      d = fmax(d, 0.0);
      closest[i] = i < nvalid ? d : 0.0;
      part += closest[i];
    }
    block_sum(&part, 1);
    pot = pots[0];
    __syncthreads();
  }
  int rpos = 0;
  for (int c = 1; c < k; ++c) {
    if (tid < trials) {
      rvals[tid] = rnd[rpos + tid] * pot;
      cand[tid] = n - 1;  // np.clip(candidate_ids, None, n - 1)
    }
    rpos += trials;
    // cumulative sum in row order: running sum inside the thread + scan of thread totals
    double mysum = 0.0;
#pragma unroll
    for (int i = 0; i < NR; ++i) mysum += closest[i];
    {
      double v = mysum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o);
        if (lane >= o) v += u;
      }
      if (lane == 63) sm[wave] = v;
      __syncthreads();
      double off = 0.0;
      for (int w = 0; w < wave; ++w) off += sm[w];
      scan[tid] = v + off;
    }
    __syncthreads();
    {
      const double excl = tid == 0 ? 0.0 : scan[tid - 1];
      const double incl = scan[tid];
      for (int t = 0; t < trials; ++t) {
        const double rv = rvals[t];
        // searchsorted(cumsum, rv, 'left'): first index with cumsum >= rv
        if (nvalid > 0 && (rv > excl || tid == 0) && rv <= incl) {
          double run = excl;
          int hit = r0 + nvalid - 1;
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            run += closest[i];
            if (i < nvalid - 1 && run >= rv && hit == r0 + nvalid - 1) hit = r0 + i;
          }
          atomicMin(&cand[t], hit);
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < trials * k; e += NT) {
      const int t = e / k, j = e - t * k;
      candrow[t * k + j] = ET[(size_t)j * lde + cand[t]] - mean[j];
    }
    __syncthreads();
    if (tid < trials) {
      double s2 = 0.0;
      for (int j = 0; j < k; ++j) s2 += candrow[tid * k + j] * candrow[tid * k + j];
      candsq[tid] = s2;
    }
    __syncthreads();
    // trials two at a time (register budget: 1024 threads -> 128 VGPRs each)
    double best_pot = INFINITY;
    int best_t = 0;
    for (int t0 = 0; t0 < trials; t0 += 2) {
      double dot[2][NR];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < NR; ++i) dot[t][i] = 0.0;
      for (int j = 0; j < k; ++j) {
        double v[NR];
        load_rows<NR>(ET + (size_t)j * lde, r0, n, full, v);
        const double mj = mean[j];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const double cj = (t0 + t < trials) ? candrow[(t0 + t) * k + j] : 0.0;
#pragma unroll
          for (int i = 0; i < NR; ++i) dot[t][i] += cj * (v[i] - mj);
        }
      }
      double part[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        part[t] = 0.0;
        const double csq = (t0 + t < trials) ? candsq[t0 + t] : 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          double d = -2.0 * dot[t][i];
          d += csq;
          d += xsq[i];
          d = fmax(d, 0.0);
          d = fmin(closest[i], d);
          part[t] += i < nvalid ? d : 0.0;
        }
      }
      block_sum(part, 2);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t0 + t < trials && pots[t] < best_pot) {  // np.argmin: first minimum
          best_pot = pots[t];
          best_t = t0 + t;
        }
      }
      __syncthreads();
    }
    // closest <- min(closest, distance to the winning candidate): recomputed (one more
    // pass) instead of keeping every trial's distances alive in registers
    {
      double dot[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) dot[i] = 0.0;
      for (int j = 0; j < k; ++j) {
        double v[NR];
        load_rows<NR>(ET + (size_t)j * lde, r0, n, full, v);
        const double mj = mean[j], cj = candrow[best_t * k + j];
#pragma unroll
        for (int i = 0; i < NR; ++i) dot[i] += cj * (v[i] - mj);
      }
      const double csq = candsq[best_t];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        double d = -2.0 * dot[i];
        d += csq;
        d += xsq[i];
        d = fmax(d, 0.0);
        closest[i] = i < nvalid ? fmin(closest[i], d) : 0.0;
      }
    }
    pot = best_pot;
    if (tid == 0) seeds[c] = cand[best_t];
    __syncthreads();
  }
  if (tid == 0) info[2] = (int)(wall_clock64() - t_start);

  // shared: assignment of this thread's rows against `cent` (k x k).
  //   euclid = true : argmin |c|^2 - 2 x.c on centred data (cnorm = |c|^2)
  //   euclid = false: argmin 1 - clip(e.c / (|e| |c|))     (cnorm = |c|)
  int label[NR];
  double enorm[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) enorm[i] = 0.0;
  auto assign = [&](bool euclid) -> double {
    double dsum = 0.0;
#pragma unroll
    for (int h = 0; h < NR; h += 4) {
      double dot[KC][4];
#pragma unroll
      for (int c2 = 0; c2 < KC; ++c2)
#pragma unroll
        for (int i = 0; i < 4; ++i) dot[c2][i] = 0.0;
      for (int j = 0; j < k; ++j) {
        double v[NR];
        load_rows<NR>(ET + (size_t)j * lde, r0, n, full, v);
        const double mj = euclid ? mean[j] : 0.0;
#pragma unroll
        for (int c2 = 0; c2 < KC; ++c2) {
          const double cj = c2 < k ? cent[c2 * k + j] : 0.0;
#pragma unroll
          for (int i = 0; i < 4; ++i) dot[c2][i] += (v[h + i] - mj) * cj;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int best = 0;
        double bd = INFINITY;
#pragma unroll
        for (int c2 = 0; c2 < KC; ++c2) {
          if (c2 < k) {
            double d;
            if (euclid) {
              d = cnorm[c2] - 2.0 * dot[c2][i];
            } else {
              double cosine = dot[c2][i] / (enorm[h + i] * cnorm[c2]);
              if (fabs(cosine) > 1.0) cosine = copysign(1.0, cosine);
              d = 1.0 - cosine;
            }
            if (d < bd) { bd = d; best = c2; }
          }
        }
        label[h + i] = best;
        if (h + i < nvalid) dsum += bd;
      }
    }
    return dsum;
  };
  // shared: per-cluster means of the member rows, all clusters in ONE pass over the data
  //   centred = true : Lloyd (mean of x - mean, + mean; empty cluster keeps its seed)
  //   centred = false: cosine loop (`.any()` on the member indices)
  auto update = [&](bool centred) {
    int cnt[KC], nz[KC];
#pragma unroll
    for (int c2 = 0; c2 < KC; ++c2) {
      cnt[c2] = 0;
      nz[c2] = 0;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int hit = (i < nvalid) && (label[i] == c2);
        cnt[c2] += hit;
        nz[c2] += hit && (r0 + i > 0);
      }
      cnt[c2] = wsumi(cnt[c2]);
      nz[c2] = wsumi(nz[c2]);
      if (lane == 0) { wcnt[wave * KC + c2] = cnt[c2]; wnz[wave * KC + c2] = nz[c2]; }
    }
    for (int j = 0; j < k; ++j) {
      double v[NR];
      load_rows<NR>(ET + (size_t)j * lde, r0, n, full, v);
      const double mj = centred ? mean[j] : 0.0;
#pragma unroll
      for (int c2 = 0; c2 < KC; ++c2) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (i < nvalid && label[i] == c2) a += v[i] - mj;
        a = wsum(a);
        if (lane == 0) wsumv[(wave * KC + c2) * KC + j] = a;
      }
    }
    __syncthreads();
    if (tid < k * k) {
      const int c2 = tid / k, j = tid - c2 * k;
      double tot = 0.0;
      int count = 0, nzc = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        tot += wsumv[(w * KC + c2) * KC + j];
        count += wcnt[w * KC + c2];
        nzc += wnz[w * KC + c2];
      }
      if (centred) {
        const double vv = count > 0 ? tot / (double)count : cent[c2 * k + j];
        cent[c2 * k + j] = vv + mean[j];
      } else if (nzc > 0) {
        cent[c2 * k + j] = tot / (double)count;
      }
    }
    __syncthreads();
  };

  // ---- one Euclidean Lloyd step on the centred data (max_iter = 1) -----------------------
  for (int e = tid; e < k * k; e += NT) {
    const int c2 = e / k, j = e - c2 * k;
    cent[e] = ET[(size_t)j * lde + seeds[c2]] - mean[j];
  }
  __syncthreads();
  if (tid < k) {
    double s2 = 0.0;
    for (int j = 0; j < k; ++j) s2 += cent[tid * k + j] * cent[tid * k + j];
    cnorm[tid] = s2;
  }
  __syncthreads();
  assign(true);
  __syncthreads();  // every thread is done reading `cent` before update rewrites it
  update(true);
  if (tid == 0) info[3] = (int)(wall_clock64() - t_start);

  // ---- CustomKMeans.predict, cosine (custom_distance_kmeans.py:118-141) ------------------
  for (int j = 0; j < k; ++j) {
    double v[NR];
    load_rows<NR>(ET + (size_t)j * lde, r0, n, full, v);
#pragma unroll
    for (int i = 0; i < NR; ++i) enorm[i] += v[i] * v[i];
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) enorm[i] = sqrt(enorm[i]);
  double prev = 0.0;
  int it = 0;
  for (;; ++it) {
    if (tid < k) {
      double s2 = 0.0;
      for (int j = 0; j < k; ++j) s2 += cent[tid * k + j] * cent[tid * k + j];
      cnorm[tid] = sqrt(s2);
    }
    __syncthreads();
    double part = assign(false);
    block_sum(&part, 1);
    const double mean_d = pots[0] / (double)n;
    __syncthreads();
    if ((mean_d <= prev && mean_d >= (1.0 - 0.001) * prev) || it == max_iter) break;
    prev = mean_d;
    update(false);
  }
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (i < nvalid) labels64[r0 + i] = label[i];
  for (int e = tid; e < k * k; e += NT) cent_out[e] = cent[e];
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
                   const KmeansWorkspace& ws, int metric) {
  if (metric == kKmeansCosine && k <= 8 && n <= 16 * 512 && trials <= 8) {
    // register-resident fast path: 512 threads (256 VGPRs each), contiguous rows per thread
    if (n <= 8 * 512)
      hipLaunchKernelGGL((k_kmeans_fast<8, 512>), dim3(1), dim3(512), 0, s, ET, lde, n, k,
                         max_iter, first_center, trials, ws.rnd, ws.centroids, ws.labels64,
                         ws.info);
    else
      hipLaunchKernelGGL((k_kmeans_fast<16, 512>), dim3(1), dim3(512), 0, s, ET, lde, n, k,
                         max_iter, first_center, trials, ws.rnd, ws.centroids, ws.labels64,
                         ws.info);
    return;
  }
  if (k > kMaxVectors) {
    hipLaunchKernelGGL(k_kmeans<true>, dim3(1), dim3(KT), 0, s, ET, lde, n, k, max_iter,
                       first_center, trials, ws.Xc, ws.xsq, ws.closest, ws.cand, ws.enorm, ws.rnd,
                       ws.centroids, ws.labels32, ws.labels64, ws.info, metric, ws.big,
                       ws.big_words);
    return;
  }
  hipLaunchKernelGGL(k_kmeans<false>, dim3(1), dim3(KT), 0, s, ET, lde, n, k, max_iter,
                     first_center, trials, ws.Xc, ws.xsq, ws.closest, ws.cand,
                     ws.enorm, ws.rnd, ws.centroids, ws.labels32, ws.labels64,
                     ws.info, metric, static_cast<double*>(nullptr), static_cast<int*>(nullptr));
}

}  // namespace sc
