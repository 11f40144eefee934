// Host side of the callers around the spectral path: tree
// bookkeeping of the agglomerative clustering, centroids, fallback decisions, naive
// clusterer; kernels in ahc.hip and fallback.hip.
#include "handle.h"

// ------------------------------------------------------------------------------
// N4: size reduction -- agglomerative clustering + centroids
// ------------------------------------------------------------------------------
namespace {
// CPython heapq (Lib/heapq.py) on ints: sklearn's _hc_cut enumerates the heap ARRAY, so
// the exact sift order defines the label numbering.
void heap_siftdown(std::vector<long long>& heap, size_t startpos, size_t pos) {
  const long long newitem = heap[pos];
  while (pos > startpos) {
    const size_t parentpos = (pos - 1) >> 1;
    const long long parent = heap[parentpos];
    if (newitem < parent) {
      heap[pos] = parent;
      pos = parentpos;
      continue;
    }
    break;
  }
  heap[pos] = newitem;
}
void heap_siftup(std::vector<long long>& heap, size_t pos) {
  const size_t endpos = heap.size(), startpos = pos;
  const long long newitem = heap[pos];
  size_t childpos = 2 * pos + 1;
  while (childpos < endpos) {
    const size_t rightpos = childpos + 1;
    if (rightpos < endpos && !(heap[childpos] < heap[rightpos])) childpos = rightpos;
    heap[pos] = heap[childpos];
    pos = childpos;
    childpos = 2 * pos + 1;
  }
  heap[pos] = newitem;
  heap_siftdown(heap, startpos, pos);
}
void heap_push(std::vector<long long>& heap, long long item) {
  heap.push_back(item);
  heap_siftdown(heap, 0, heap.size() - 1);
}
void heap_pushpop(std::vector<long long>& heap, long long item) {
  if (!heap.empty() && heap[0] < item) {
    std::swap(item, heap[0]);
    heap_siftup(heap, 0);
  }
}
}  // namespace

// sklearn.cluster.AgglomerativeClustering(metric="cosine", linkage=complete|average,
// n_clusters=... | distance_threshold=...).fit_predict(X), label numbering included.
extern "C" int sc_ahc(sc_handle h, const double* x, int n, int d, int linkage, int n_clusters,
                      double distance_threshold, int64_t* labels, int* n_clusters_out) {
  if (!h) return SC_ERR_INVALID;
  if (!x || !labels || d <= 0) return fail(h, SC_ERR_INVALID, "embeddings must be (n, d)");
  if (n < 2)
    return fail(h, SC_ERR_INVALID,
                "Found array with " + std::to_string(std::max(n, 0)) +
                    " sample(s) while a minimum of 2 is required by AgglomerativeClustering.");
  if (linkage != SC_LINKAGE_COMPLETE && linkage != SC_LINKAGE_AVERAGE)
    return fail(h, SC_ERR_INVALID, "linkage must be complete or average");
  if (n_clusters < 0 || n_clusters > n)
    return fail(h, SC_ERR_INVALID, "Cannot extract more clusters than samples");
  SC_HIP(h, hipSetDevice(h->device));
  // cosine distances: the affinity stage's normalise + symmetric GEMM, then 1 - clip(c)
  SC_TRY(sc_set_embeddings(h, x, n, d));
  SC_TRY(ensure_tilemap(h, n));
  hipStream_t s = h->stream;
  const int ld = h->ldn;
  launch_normalize_rows(s, ptr<double>(h->X), h->ldx, n, d, ptr<double>(h->Xn));
  launch_gemm_nt(s, ptr<double>(h->Xn), h->ldx, ptr<double>(h->Xn), h->ldx, ptr<double>(h->B1),
                 ld, n, n, d, kEpiNone, true, ptr<double>(h->splitk), h->tilemap_cur);
  launch_cosine_distance(s, ptr<double>(h->B1), n, ld);
  SC_TRY(grow(h, h->ahc_size, (size_t)n * sizeof(int)));
  SC_TRY(grow(h, h->ahc_chain, (size_t)n * sizeof(int)));
  SC_TRY(grow(h, h->ahc_Z, (size_t)n * 4 * sizeof(double)));
  launch_ahc_nn_chain(s, ptr<double>(h->B1), ld, n, linkage, ptr<int>(h->ahc_size),
                      ptr<int>(h->ahc_chain), ptr<double>(h->ahc_Z));
  SC_TRY(check_last(h, "agglomerative clustering launch"));
  std::vector<double> Z((size_t)(n - 1) * 4);
  SC_HIP(h, hipMemcpyAsync(Z.data(), h->ahc_Z.p, Z.size() * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  h->have_affinity = h->have_cropval = false;
  // ---- scipy: stable sort by height, union-find relabelling (hierarchy.pyx `label`)
  std::vector<int> order(n - 1);
  for (int i = 0; i < n - 1; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(),
                   [&](int a, int b) { return Z[(size_t)a * 4 + 2] < Z[(size_t)b * 4 + 2]; });
  std::vector<int> parent(2 * (size_t)n - 1);
  for (size_t i = 0; i < parent.size(); ++i) parent[i] = (int)i;
  auto find = [&](int v) {
    int r = v;
    while (parent[r] != r) r = parent[r];
    while (parent[v] != r) {
      const int next = parent[v];
      parent[v] = r;
      v = next;
    }
    return r;
  };
  std::vector<std::array<long long, 2>> children(n - 1);
  std::vector<double> heights(n - 1);
  int next_label = n;
  for (int i = 0; i < n - 1; ++i) {
    const int m = order[i];
    const int xr = find((int)Z[(size_t)m * 4]), yr = find((int)Z[(size_t)m * 4 + 1]);
    children[i] = {std::min(xr, yr), std::max(xr, yr)};
    heights[i] = Z[(size_t)m * 4 + 2];
    parent[xr] = next_label;
    parent[yr] = next_label;
    ++next_label;
  }
  // ---- sklearn: number of clusters, then _hc_cut
  int k = n_clusters;
  if (k == 0) {  // distance_threshold mode
    k = 1;
    for (int i = 0; i < n - 1; ++i) k += heights[i] >= distance_threshold;
  }
  if (n_clusters_out) *n_clusters_out = k;
  std::vector<long long> nodes;
  nodes.push_back(-(std::max(children[n - 2][0], children[n - 2][1]) + 1));
  for (int it = 0; it < k - 1; ++it) {
    const auto c = children[(size_t)(-nodes[0] - n)];
    heap_push(nodes, -c[0]);
    heap_pushpop(nodes, -c[1]);
  }
  std::vector<long long> stack;
  for (size_t i = 0; i < nodes.size(); ++i) {
    stack.assign(1, -nodes[i]);
    while (!stack.empty()) {
      const long long v = stack.back();
      stack.pop_back();
      if (v < n) {
        labels[v] = (int64_t)i;
      } else {
        stack.push_back(children[(size_t)(v - n)][0]);
        stack.push_back(children[(size_t)(v - n)][1]);
      }
    }
  }
  return SC_OK;
}

// utils.get_cluster_centroids (reference utils.py:159-176): (k, d) means, k = max(labels)+1
extern "C" int sc_cluster_centroids(sc_handle h, const double* x, int n, int d,
                                    const int64_t* labels, int k, double* out) {
  if (!h) return SC_ERR_INVALID;
  if (!x || !labels || !out || n <= 0 || d <= 0 || k <= 0)
    return fail(h, SC_ERR_INVALID, "embeddings must be (n, d), labels (n,)");
  SC_HIP(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  SC_TRY(grow(h, h->ahc_lab, (size_t)n * sizeof(int)));
  SC_TRY(grow(h, h->ahc_cent, (size_t)(n + k) * d * sizeof(double)));
  std::vector<int> lab32(n);
  for (int i = 0; i < n; ++i) lab32[i] = (int)labels[i];
  double* xd = ptr<double>(h->ahc_cent);
  double* cd_ = xd + (size_t)n * d;
  SC_HIP(h, hipMemcpyAsync(xd, x, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, s));
  SC_HIP(h, hipMemcpyAsync(h->ahc_lab.p, lab32.data(), (size_t)n * sizeof(int),
                           hipMemcpyHostToDevice, s));
  launch_cluster_centroids(s, xd, d, n, d, ptr<int>(h->ahc_lab), k, cd_);
  SC_TRY(check_last(h, "centroid launch"));
  SC_HIP(h, hipMemcpyAsync(out, cd_, (size_t)k * d * sizeof(double), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  return SC_OK;
}

// ------------------------------------------------------------------------------
// N4: fallback decisions (reference fallback_clusterer.py, naive_clusterer.py)
// ------------------------------------------------------------------------------
// out = {affinity.min(), np.diag(affinity, k=1).min(), mean, np.std(affinity)} of the
// resident affinity (single-cluster conditions AllAffinity / NeighborAffinity / AffinityStd)
extern "C" int sc_affinity_stats(sc_handle h, double* out) {
  if (!h) return SC_ERR_INVALID;
  if (!out) return fail(h, SC_ERR_INVALID, "out is NULL");
  if (!h->have_affinity) return fail(h, SC_ERR_INVALID, "no affinity resident");
  SC_HIP(h, hipSetDevice(h->device));
  const int n = h->n;
  SC_TRY(grow(h, h->fb_part, (size_t)n * 8 * sizeof(double)));
  SC_TRY(grow(h, h->fb_small, 32 * sizeof(double)));
  launch_affinity_stats(h->stream, ptr<double>(h->A0), n, h->ldn, ptr<double>(h->fb_part),
                        ptr<double>(h->fb_small));
  SC_TRY(check_last(h, "affinity statistics launch"));
  SC_HIP(h, hipMemcpyAsync(out, h->fb_small.p, 4 * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  return SC_OK;
}

// BIC of a 1- and a 2-component Gaussian mixture fitted to affinity[i][j], j >= i + offset
// (fallback_clusterer.py:154-173).  sklearn's GaussianMixture defaults: full covariance,
// reg_covar 1e-6, tol 1e-3 on the mean log-likelihood, max_iter 100, k-means start.  The
// reference's k-means start is randomly seeded; here it is the deterministic 1-D 2-means
// from (min, max), which is the fixed point those seeds reach on separable data.
extern "C" int sc_affinity_gmm_bic(sc_handle h, int diagonal_offset, double* bic1,
                                   double* bic2) {
  if (!h) return SC_ERR_INVALID;
  if (!bic1 || !bic2) return fail(h, SC_ERR_INVALID, "NULL output");
  if (!h->have_affinity) return fail(h, SC_ERR_INVALID, "no affinity resident");
  const int n = h->n;
  if (diagonal_offset < 0 || diagonal_offset >= n - 1)
    return fail(h, SC_ERR_INVALID,
                "single_cluster_affinity_diagonal_offset must be significantly smaller than "
                "affinity matrix dimension");
  SC_HIP(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  SC_TRY(grow(h, h->fb_part, (size_t)n * 8 * sizeof(double)));
  SC_TRY(grow(h, h->fb_small, 32 * sizeof(double)));
  double* params_d = ptr<double>(h->fb_small);
  double* sums_d = params_d + 8;
  const double* a = ptr<double>(h->A0);
  const int ld = h->ldn;
  const double m = (double)(n - diagonal_offset);
  const double count = m * (m + 1.0) / 2.0;
  const double reg = 1e-6, tiny = 10.0 * 2.220446049250313e-16;
  double sums[8];
  auto pass = [&](int components, int mode, const double* params) -> int {
    SC_HIP(h, hipMemcpyAsync(params_d, params, 6 * sizeof(double), hipMemcpyHostToDevice, s));
    launch_gmm_pass(s, a, n, ld, diagonal_offset, components, mode, params_d,
                    ptr<double>(h->fb_part), sums_d);
    SC_HIP(h, hipMemcpyAsync(sums, sums_d, 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipStreamSynchronize(s));
    return SC_OK;
  };
  // M-step of sklearn's _estimate_gaussian_parameters from the pass sums
  auto m_step = [&](int components, double* params) {
    double wsum = 0.0;
    for (int c = 0; c < components; ++c) {
      const double nk = sums[3 * c] + tiny;
      const double mu = sums[3 * c + 1] / nk;
      const double var = (sums[3 * c + 2] - 2.0 * mu * sums[3 * c + 1] + mu * mu * sums[3 * c]) / nk;
      params[3 * c] = nk / count;
      params[3 * c + 1] = mu;
      params[3 * c + 2] = var + reg;
      wsum += params[3 * c];
    }
    for (int c = 0; c < components; ++c) params[3 * c] /= wsum;
  };
  auto fit = [&](int components, double* bic) -> int {
    double params[6] = {1.0, 0.0, 1.0, 0.0, 0.0, 1.0};
    if (components == 1) {
      SC_TRY(pass(1, 1, params));  // r0 = 1 everywhere: plain moments
      m_step(1, params);
    } else {
      // 2-means start from the extremes, Lloyd steps until the inertia stops moving
      launch_gmm_range(s, a, n, ld, diagonal_offset, ptr<double>(h->fb_part), sums_d);
      double range[2];
      SC_HIP(h, hipMemcpyAsync(range, sums_d, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
      SC_HIP(h, hipStreamSynchronize(s));
      params[1] = range[0];
      params[4] = range[1];
      double prev_inertia = -1.0;
      for (int it = 0; it < 300; ++it) {
        SC_TRY(pass(2, 0, params));
        const double inertia = sums[6];
        if (sums[0] > 0.0) params[1] = sums[1] / sums[0];
        if (sums[3] > 0.0) params[4] = sums[4] / sums[3];
        if (inertia == prev_inertia) break;
        prev_inertia = inertia;
      }
      SC_TRY(pass(2, 0, params));  // responsibilities = the final hard labels
      m_step(2, params);
    }
    double prev = -__builtin_huge_val();
    for (int it = 0; it < 100; ++it) {
      SC_TRY(pass(components, 1, params));  // E-step under params (+ sums of the M-step)
      const double lower_bound = sums[6] / count;
      m_step(components, params);
      if (std::fabs(lower_bound - prev) < 1e-3) break;
      prev = lower_bound;
    }
    SC_TRY(pass(components, 1, params));
    const double n_params = components == 1 ? 2.0 : 5.0;
    *bic = -2.0 * sums[6] + n_params * std::log(count);
    return SC_OK;
  };
  SC_TRY(fit(1, bic1));
  SC_TRY(fit(2, bic2));
  return SC_OK;
}

// NaiveClusterer.predict (naive_clusterer.py:57-105) continuing from the given state:
// centroids (capacity x d, the first *n_centroids rows valid), counts, labels out.
extern "C" int sc_naive_cluster(sc_handle h, const double* x, int n, int d, double threshold,
                                double adaptation_threshold, double* centroids, int32_t* counts,
                                int32_t* n_centroids, int capacity, int64_t* labels) {
  if (!h) return SC_ERR_INVALID;
  if (!x || !centroids || !counts || !n_centroids || !labels || n <= 0 || d <= 0)
    return fail(h, SC_ERR_INVALID, "embeddings must be (n, d)");
  if (*n_centroids < 0 || *n_centroids + n > capacity)
    return fail(h, SC_ERR_INVALID, "centroid capacity must cover n_centroids + n");
  SC_HIP(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  SC_TRY(grow(h, h->fb_x, (size_t)n * d * sizeof(double)));
  SC_TRY(grow(h, h->fb_cent, (size_t)capacity * d * sizeof(double)));
  SC_TRY(grow(h, h->fb_int, ((size_t)capacity + n + 4) * sizeof(int)));
  int* counts_d = ptr<int>(h->fb_int);
  int* k_d = counts_d + capacity;
  int* labels_d = k_d + 4;
  const int k0 = *n_centroids;
  SC_HIP(h, hipMemcpyAsync(h->fb_x.p, x, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, s));
  if (k0 > 0) {
    SC_HIP(h, hipMemcpyAsync(h->fb_cent.p, centroids, (size_t)k0 * d * sizeof(double),
                             hipMemcpyHostToDevice, s));
    SC_HIP(h, hipMemcpyAsync(counts_d, counts, (size_t)k0 * sizeof(int), hipMemcpyHostToDevice, s));
  }
  SC_HIP(h, hipMemcpyAsync(k_d, n_centroids, sizeof(int), hipMemcpyHostToDevice, s));
  launch_naive_cluster(s, ptr<double>(h->fb_x), n, d, threshold, adaptation_threshold,
                       ptr<double>(h->fb_cent), counts_d, k_d, labels_d);
  SC_TRY(check_last(h, "naive clusterer launch"));
  std::vector<int> lab(n);
  SC_HIP(h, hipMemcpyAsync(lab.data(), labels_d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(n_centroids, k_d, sizeof(int), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  const int k1 = *n_centroids;
  SC_HIP(h, hipMemcpyAsync(centroids, h->fb_cent.p, (size_t)k1 * d * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(counts, counts_d, (size_t)k1 * sizeof(int), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) labels[i] = lab[i];
  return SC_OK;
}

