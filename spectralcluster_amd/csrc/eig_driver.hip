// Host-side control loops of the two eigen paths: block Lanczos (symmetric operator) and
// block Arnoldi (general matrix) -- basis growth, full re-orthogonalisation, Rayleigh-Ritz,
// the convergence analysis that replays the eigengap rule, restarts.  Every O(n) or larger
// computation is a kernel of eig.hip / eig_general.hip.
#include <ctime>

#include "handle.h"
#include "host_eig.h"
#include "host_pool.h"

// ------------------------------------------------------------------------------
// symmetric top-k eigensolver driver
// ------------------------------------------------------------------------------

// How far the eigenvalue that Ritz value i approximates can be from theta[i] (descending Ritz
// values of a SYMMETRIC operator, residual norms resid[]).  Always <= resid[i]; and when the
// neighbouring Ritz values are themselves located well enough to fence theta[i] off -- every
// other eigenvalue at least delta away -- the Kato-Temple bound resid^2 / delta, which is what
// makes a clustered bulk affordable: neighbours 3e-4 apart are told apart at residual 1e-5,
// where the linear bound asks for 1e-6 and a 128-vector basis has to restart to get there.
// (Both bounds say "an eigenvalue lies this close"; neither can see an eigenvalue the Krylov
// space has missed altogether -- the block of 8 start vectors guards against that, and
// block_multiplicity_suspect() below catches the one case a block of 8 cannot: an eigenvalue of
// multiplicity > 8 in front of the decisive gap.)
// The Kato-Temple bound is a HEURISTIC tightening: its gap delta comes from Ritz values and
// residual estimates, not from the spectrum, and the factor 2 below is empirical.  What is
// PROVEN about an accepted value of a SYMMETRIC operator is the plain residual bound, which
// analyze() caps at max(value_tol, 1e-5) |theta|: |lambda - theta| <= resid <= 1e-5 |theta| for
// every consumed value whenever value_tol <= 1e-5 (the default is 1e-6), whatever Kato-Temple
// says.  (General path: every consumed value is held to value_tol by its residual since round 6;
// there the residual bounds the error only up to the eigenvalue's condition number.)
static double value_error_bound(const double* theta, const double* resid, int m, int i,
                                bool symmetric) {
  const double r = resid[i];
  if (!symmetric || !(r > 0.0) || !std::isfinite(r)) return r;
  double delta = __builtin_huge_val();
  if (i > 0) delta = std::min(delta, theta[i - 1] - resid[i - 1] - theta[i]);
  if (i + 1 < m) delta = std::min(delta, theta[i] - theta[i + 1] - resid[i + 1]);
  else return r;  // the last Ritz value has nothing below it to fence it off
  delta -= r;
  if (!(delta > 0.0) || !std::isfinite(delta)) return r;
  // (x 2: delta comes from Ritz values and residual ESTIMATES, not from the spectrum itself; one
  //  of 160 fuzz cases sat at 1.2 x the bound without it, and the factor costs nothing that
  //  config 4 or 5 can measure)
  return std::min(r, 2.0 * r * r / delta);
}

extern "C" int sc_host_value_error_bound(const double* theta, const double* resid, int m, int i,
                                         double* bound) {
  if (!theta || !resid || !bound || m < 1 || i < 0 || i >= m) return SC_ERR_INVALID;
  *bound = value_error_bound(theta, resid, m, i, true);
  return SC_OK;
}

// Inspect Ritz values theta[0..m) (descending) + residual estimates.
static EigDecision analyze(const EigRequest& rq, const double* theta, const double* resid,
                           int m, int n, bool exact, bool symmetric_op = true) {
  EigDecision dc;
  std::vector<double> w(m);
  for (int i = 0; i < m; ++i) w[i] = rq.descend ? theta[i] : -theta[i];
  const double scale = std::max(std::fabs(theta[0]), std::fabs(theta[m - 1]));
  int kw;
  if (rq.fixed_count > 0) {
    kw = std::min(rq.fixed_count, n);
  } else if (rq.max_clusters > 0) {
    kw = std::min(n, rq.max_clusters + 1);
  } else if (rq.descend) {
    // max_clusters None: everything >= stop_eigenvalue, plus the first one below
    int c = 0;
    while (c < m && !(w[c] < rq.stop_eigenvalue)) ++c;
    if (c >= m && m < n) {  // have not reached the stop value yet
      // Ritz values never exceed the eigenvalues they approximate: m of them above the stop
      // value mean more than m are read -- beyond half the basis cap that is a job for the
      // dense full-spectrum path
      if (m >= kEigBasisCap / 2 && !exact) dc.unsupported = true;
      return dc;
    }
    kw = std::min(c + 1, n);
  } else {
    kw = n;  // ascending without max_clusters reads every eigenvalue
  }
  if (kw > m) {
    if (kw > kEigBasisCap / 2 && !exact) dc.unsupported = true;
    return dc;
  }
  dc.enough = true;
  dc.kw = kw;
  if (rq.fixed_count > 0) {
    dc.kvec = kw;
  } else {
    // np.max(eigenvalues) is taken over the WHOLE spectrum (utils.py:110,123): the
    // first value when descending, the far end of the Ritz spectrum when ascending.
    // (general path, ascending NormalizedDiff: the far end from its own solve, gen_topk)
    const double wmax = rq.have_far ? rq.far_value : (rq.descend ? w[0] : w[m - 1]);
    eigengap_core(w.data(), kw, rq.max_clusters, rq.use_stop ? rq.stop_eigenvalue : 0.0,
                  rq.eigengap_type, rq.descend, wmax, &dc.n_clusters_raw, &dc.max_delta);
    dc.kvec = std::max(dc.n_clusters_raw, rq.min_clusters);
    if (dc.kvec < 1) dc.kvec = 1;
    if (dc.kvec > kEigBasisCap / 2 && !exact) {  // (min_clusters beyond what a basis holds)
      dc.enough = false;
      dc.unsupported = true;
      return dc;
    }
    if (dc.kvec > m) dc.kvec = std::min(m, n);
  }
  if (exact) {
    dc.converged = true;
    return dc;
  }
  bool ok = true;
  const double floor_abs = 1e-14 * scale;
  // values actually read by the eigengap loop
  int first = rq.descend ? 0 : 1, last = kw - 1;
  if (rq.fixed_count > 0) first = 0;
  if (rq.descend && rq.use_stop && rq.fixed_count == 0) {
    for (int i = 0; i < kw; ++i)
      if (w[i] < rq.stop_eigenvalue) { last = i; break; }
  }
  const bool aware = rq.decision_aware && rq.fixed_count == 0;
  const int kb = dc.n_clusters_raw;  // the maximum gap sits between w[kb - 1] and w[kb]
  for (int i = first; i <= last; ++i) {
    const bool decisive = !aware || i == kb - 1 || i == kb ||
                          (rq.eigengap_type == SC_EIGENGAP_NORMALIZED_DIFF && rq.descend && i == 0);
    // (round 6: EVERY consumed value is held to value_tol -- the parity bar is on all of them.
    //  Rounds 3-5 held the non-decisive ones to 1e-3 only; SC_GEN_LOOSE_BULK=1 brings that
    //  back for the pass-count A/B of profiles/r18.  Decision-awareness survives as the
    //  ADDITIONAL interval proof below.)
    const double rel = (decisive || !sw::gen_loose_bulk()) ? rq.value_tol
                                                            : std::max(rq.value_tol, 1e-3);
    const double tol = std::max(rel * std::fabs(w[i]), floor_abs);
    // (the proven part: the residual itself within the parity bar of 1e-5 -- the Kato-Temple
    //  estimate may accept a residual above `tol`, never one above this)
    //  (measured, profiles/r07b_passes_probe.txt: 2176 instead of 2115 block passes on config 5's
    //  512 utterances, +2.9 %, throughput inside the run-to-run spread; none on configs 3 and 4)
    const double cap = std::max(std::max(rel, 1e-5) * std::fabs(w[i]), floor_abs);
    if (!(value_error_bound(theta, resid, m, i, symmetric_op) <= tol) || !(resid[i] <= cap)) {
      if (ok) { dc.fail_kind = 1; dc.fail_index = i; }
      ok = false;
    }
    dc.max_resid = std::max(dc.max_resid, resid[i]);
  }
  if (aware && ok) {
    // interval check: eigenvalue i lies within err(i) of w[i] (10 x residual: a safety
    // factor for the departure from normality)
    auto err = [&](int i) { return 10.0 * resid[i]; };
    const double eps = 1e-10;
    const double wmax = rq.have_far ? rq.far_value : (rq.descend ? w[0] : w[m - 1]);
    auto gap_bounds = [&](int lo_i, int hi_i, double* lower, double* upper) {
      // Ratio: w[hi_i] / (w[lo_i] + eps); NormalizedDiff: (w[hi_i] - w[lo_i]) / wmax,
      // where hi_i is the numerator index
      const double a = w[hi_i], b = w[lo_i], ea = err(hi_i), eb = err(lo_i);
      if (rq.eigengap_type == SC_EIGENGAP_RATIO) {
        const double den_lo = b - eb + eps, den_hi = b + eb + eps;
        *upper = den_lo > 0.0 ? (a + ea) / den_lo : 1e300;
        *lower = den_hi > 0.0 ? (a - ea) / den_hi : -1e300;
      } else {
        *upper = (a - b + ea + eb) / wmax;
        *lower = (a - b - ea - eb) / wmax;
      }
    };
    double best_lo = 0.0, dummy;
    if (kb >= 1) {
      if (rq.descend) gap_bounds(kb, kb - 1, &best_lo, &dummy);
      else gap_bounds(kb - 1, kb, &best_lo, &dummy);
    }
    const int end = kw;
    if (rq.descend) {
      for (int i = 1; i < end && ok; ++i) {
        if (rq.use_stop) {
          if (std::fabs(w[i - 1] - rq.stop_eigenvalue) <= err(i - 1)) {
            ok = false; dc.fail_kind = 4; dc.fail_index = i - 1;
            break;
          }
          if (w[i - 1] < rq.stop_eigenvalue) break;
        }
        if (i == kb) continue;
        double lo, up;
        gap_bounds(i, i - 1, &lo, &up);
        if (!(up < best_lo) && !(kb == 0 && up <= 0.0)) {
          ok = false; dc.fail_kind = 4; dc.fail_index = i;
        }
      }
    } else {
      for (int i = 1; i < end - 1 && ok; ++i) {
        if (i + 1 == kb) continue;
        double lo, up;
        gap_bounds(i, i + 1, &lo, &up);
        if (!(up < best_lo) && !(kb == 0 && up <= 0.0)) {
          ok = false; dc.fail_kind = 4; dc.fail_index = i;
        }
      }
    }
  }
  // (np.max(eigenvalues) of the ascending NormalizedDiff branch is the far end of the
  //  spectrum, on the edge of a dense bulk where Krylov methods converge like 1 / degree^2:
  //  that request never reaches this loop -- sym_topk takes it from the dense path)
  for (int i = 0; i < dc.kvec; ++i) {
    if (!(resid[i] <= std::max(rq.vector_tol * scale, floor_abs))) {
      if (ok) { dc.fail_kind = 3; dc.fail_index = i; }
      ok = false;
    }
    dc.max_resid = std::max(dc.max_resid, resid[i]);
  }
  dc.converged = ok;
  return dc;
}

// A block Krylov space built from 8 start vectors holds at most 8 independent vectors of any one
// eigenspace: of an eigenvalue of multiplicity > 8 (or a cluster tighter than the stopping
// tolerance) it finds exactly 8 copies -- all converged, all with tiny residuals -- until rounding
// noise has grown the others, which a fast-converging request never waits for.  No residual
// bound sees the missing copies.  The signature is unmistakable, though: kEigBlock consumed Ritz
// values equal within 10 x value_tol.  It only matters where missing copies would shift what the
// caller reads: in front of the decisive gap (between w[kb - 1] and w[kb]) for an eigengap
// request, in front of the last requested value for a fixed count.  (Behind the gap, equal
// values give ratios of 1 / differences of 0 however many there are: the bulk at 1.0 of a
// GraphCut Laplacian, 13 of the 21 values read at n = 8192, is not a suspect.)  A suspect
// spectrum goes to the dense path, which counts eigenvalues exactly (Sturm sequences).
static bool block_multiplicity_suspect(const EigRequest& rq, const EigDecision& dc,
                                       const double* theta, int m) {
  const int kw = std::min(dc.kw, m);
  const int limit = rq.fixed_count > 0 ? kw - 2 : dc.n_clusters_raw - 1;  // last index of a run that matters
  const double rel = 10.0 * std::max(rq.value_tol, 1e-12);
  const double scale = std::max(std::fabs(theta[0]), std::fabs(theta[m - 1]));
  for (int i = 0; i + kEigBlock - 1 <= limit && i + kEigBlock - 1 < kw; ++i) {
    const double a = theta[i], b = theta[i + kEigBlock - 1];
    if (std::fabs(a - b) <= rel * std::max(std::max(std::fabs(a), std::fabs(b)), 1e-10 * scale))
      return true;
  }
  return false;
}

// T (m x m, from the device, row-major ld) and the residual block's Gram G (B x B) ->
// theta descending, resid estimates sqrt(y_last^T G y_last), Y (m x m, row-major ldy,
// column `rank` = Ritz vector of theta[rank]).  Same outputs as k_jacobi.
static bool host_rayleigh_ritz(const double* T, int ld, const double* G, int m, double* theta,
                               double* resid, double* Y, int ldy) {
  // (an odd row stride: tql2 walks columns, and a stride of 64 or 128 doubles maps a whole
  //  column onto a handful of cache sets)
  const int lda = m | 1;
  std::vector<double> a((size_t)m * lda), d(m), e(m);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j)  // the mirrored upper triangle is what the chain wrote
      a[(size_t)i * lda + j] = i <= j ? T[(size_t)i * ld + j] : T[(size_t)j * ld + i];
  if (!host_symmetric_eig(a.data(), lda, m, d.data(), e.data())) return false;
  std::vector<int> order(m);
  for (int i = 0; i < m; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d[x] > d[y]; });
  for (int rank = 0; rank < m; ++rank) {
    const int c = order[rank];
    theta[rank] = d[c];
    double r2 = 0.0;
    for (int p = 0; p < kEigBlock; ++p) {
      double t = 0.0;
      for (int q = 0; q < kEigBlock; ++q)
        t += G[p * kEigBlock + q] * a[(size_t)(m - kEigBlock + q) * lda + c];
      r2 += a[(size_t)(m - kEigBlock + p) * lda + c] * t;
    }
    resid[rank] = std::sqrt(std::max(r2, 0.0));
    for (int r = 0; r < m; ++r) Y[(size_t)r * ldy + rank] = a[(size_t)r * lda + c];
  }
  return true;
}

// The same for a basis of more than kHostRR vectors (clustered spectra, restarts): every Ritz
// value, but only the leading Ritz VECTORS -- those the analysis reads residuals of and a
// thick restart keeps (host_eig.cpp: 4/3 m^3 + O(need m^2) instead of ~9 m^3).  `count_fn`
// maps the Ritz values to the number of leading pairs the caller will look at.  Columns
// >= need of Y are zero, their residual estimates infinite.
template <typename CountFn>
static bool host_rayleigh_ritz_leading(const double* T, int ld, const double* G, int m,
                                       double* theta, double* resid, double* Y, int ldy,
                                       CountFn count_fn) {
  HostTridiag tw;
  if (!host_partial_values(T, ld, m, &tw)) return false;
  for (int i = 0; i < m; ++i) theta[i] = tw.theta[i];
  const int need = std::max(1, std::min(m, count_fn(theta)));
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < m; ++c) Y[(size_t)r * ldy + c] = 0.0;
  if (!host_partial_vectors(tw, need, Y, ldy)) return false;
  for (int rank = 0; rank < m; ++rank) {
    if (rank >= need) {
      resid[rank] = __builtin_huge_val();
      continue;
    }
    double r2 = 0.0;
    for (int p = 0; p < kEigBlock; ++p) {
      double t = 0.0;
      for (int q = 0; q < kEigBlock; ++q)
        t += G[p * kEigBlock + q] * Y[(size_t)(m - kEigBlock + q) * ldy + rank];
      r2 += Y[(size_t)(m - kEigBlock + p) * ldy + rank] * t;
    }
    resid[rank] = std::sqrt(std::max(r2, 0.0));
  }
  return true;
}

// One CholQR pass on W (n x 8) with the orthonormality-defect flag of its input armed
// (flags[10]); stores the result into Q[:, store_col ...] and Vs when store_col >= 0.
static int cholqr_pass(sc_handle h, int n, int store_col) {
  hipStream_t s = h->stream;
  double* W = ptr<double>(h->W);
  launch_proj_partial(s, W, kEigBlock, kEigBlock, W, n, ptr<double>(h->partial));
  launch_reduce_chol(s, ptr<double>(h->partial), proj_blocks(n), ptr<double>(h->Rinv), nullptr,
                     nullptr, ptr<int>(h->flags), ptr<int>(h->flags) + 10, 2);
  launch_apply_rinv(s, W, n, ptr<double>(h->Rinv), store_col >= 0 ? ptr<double>(h->Q) : nullptr,
                    kLdq, store_col >= 0 ? store_col : 0,
                    h->vs_scale ? h->vs_scale : ptr<double>(h->cvec), ptr<double>(h->Vs));
  return SC_OK;
}

static const char kNonFiniteMessage[] = "Array must not contain infs or NaNs";

// Orthonormalise W (n x 16) against Q[:, 0:m] and within itself.
//   record: accumulate the projection coefficients into T columns [col0, col0+16)
//   store_col: column of Q to receive the result (< 0: do not store)
static int orthonormalize(sc_handle h, int n, int m, bool record, int col0, int store_col,
                          bool save_gram) {
  hipStream_t s = h->stream;
  double* Q = ptr<double>(h->Q);
  double* W = ptr<double>(h->W);
  double* part = ptr<double>(h->partial);
  double* hsq = ptr<double>(h->hsq);
  SC_HIP(h, hipMemsetAsync(hsq, 0, 16 * sizeof(double), s));
  if (m > 0) {
    for (int pass = 0; pass < 2; ++pass) {
      launch_proj_partial(s, Q, kLdq, m, W, n, part);
      launch_reduce_H(s, part, proj_blocks(n), m, ptr<double>(h->Hbuf),
                      record ? ptr<double>(h->T) : nullptr, kLdq, col0, pass, hsq);
      launch_update_block(s, Q, kLdq, m, ptr<double>(h->Hbuf), W, n);
    }
  }
  // CholQR2
  launch_proj_partial(s, W, kEigBlock, kEigBlock, W, n, part);
  launch_reduce_chol(s, part, proj_blocks(n), ptr<double>(h->Rinv),
                     save_gram ? ptr<double>(h->G) : nullptr, hsq, ptr<int>(h->flags),
                     ptr<int>(h->flags) + 11, 1);
  launch_apply_rinv(s, W, n, ptr<double>(h->Rinv), nullptr, 0, 0, nullptr, nullptr);
  SC_TRY(cholqr_pass(h, n, store_col));
  return check_last(h, "orthonormalize launch");
}

static int read_flags(sc_handle h, int* mask) {
  SC_HIP(h, hipMemcpyAsync(h->h_flags, h->flags.p, 16 * sizeof(int), hipMemcpyDeviceToHost,
                           h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  *mask = h->h_flags[0];
  if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
  if (h->h_flags[1] > 0 && sw::eig_trace()) {
    fprintf(stderr, "[sc] jacobi sweeps=%d  %.1f us  %.0f MHz shader clock\n", h->h_flags[1],
            h->h_flags[2] * 0.01, h->h_flags[3] * 1024.0 / (h->h_flags[2] * 0.01));
    fprintf(stderr, "[sc]   thread-0 kcycles: param %d  barrier1 %d  update %d  barrier2 %d\n",
            h->h_flags[4], h->h_flags[5], h->h_flags[6], h->h_flags[7]);
    hipMemsetAsync(ptr<int>(h->flags) + 1, 0, 2 * sizeof(int), h->stream);
  }
  return SC_OK;
}

// Make sure the block in W is a full-rank orthonormal block; repairs dependent
// columns with random vectors (bounded retries).
static int finish_block(sc_handle h, int n, int m, int store_col, uint64_t* seed) {
  for (int attempt = 0; attempt < 4; ++attempt) {
    int mask = 0;
    SC_TRY(read_flags(h, &mask));
    // CholQR2 only orthonormalises blocks of condition < ~1e8; a numerically low-rank
    // operator produces worse ones: keep passing until the input Gram matrix was near I
    for (int extra = 0;
         extra < 3 && mask == 0 && (h->h_flags[10] != 0 || h->h_flags[11] != 0); ++extra) {
      // (the projection coefficients already recorded in T stay: this round only removes
      // rounding-level components)
      SC_TRY(orthonormalize(h, n, m, false, 0, store_col, false));
      SC_TRY(read_flags(h, &mask));
    }
    if (mask == 0) return SC_OK;
    launch_refill_deficient(h->stream, ptr<double>(h->W), n, ptr<int>(h->flags), ++(*seed));
    SC_TRY(orthonormalize(h, n, m, false, 0, store_col, false));
  }
  return fail(h, SC_ERR_NOT_CONVERGED, "could not build a full-rank Krylov block");
}

static void back_transform_cols(sc_handle h, int n, int cols) {
  launch_back_transform(h->stream, ptr<double>(h->E), round_up(n, 16), n, cols,
                        ptr<double>(h->tvec));
}

// Every eigenvalue of Op = diag(p) + diag(c) S diag(c), descending, into h->spectrum
// (eig_dense.hip: Householder tridiagonalisation + Sturm bisection).  `scratch` (n x ld)
// receives the materialised operator and is destroyed; S is left untouched.
static int dense_spectrum(sc_handle h, const double* S, int ld, int n, double* scratch) {
  hipStream_t s = h->stream;
  if (scratch == nullptr || scratch == S)
    return fail(h, SC_ERR_UNSUPPORTED, "no scratch matrix for the dense eigenvalue path");
  SC_TRY(grow(h, h->td_d, (size_t)n * sizeof(double)));
  SC_TRY(grow(h, h->td_e, (size_t)n * sizeof(double)));
  SC_TRY(grow(h, h->td_theta, (size_t)n * sizeof(double)));
  SC_TRY(grow(h, h->td_work, (size_t)(5 * (size_t)n + 2048) * sizeof(double)));
  SC_TRY(grow(h, h->td_tau, (size_t)n * sizeof(double)));
  SC_TRY(grow(h, h->td_panel, (size_t)n * 128 * sizeof(double)));
  launch_td_materialize(s, S, ld, n, ptr<double>(h->cvec), ptr<double>(h->pvec), scratch);
  launch_tridiagonalize_blocked(s, scratch, ld, n, ptr<double>(h->td_d), ptr<double>(h->td_e),
                                ptr<double>(h->td_tau), ptr<double>(h->td_panel),
                                ptr<double>(h->td_work), ptr<double>(h->splitk));
  launch_tridiagonal_eigenvalues(s, ptr<double>(h->td_d), ptr<double>(h->td_e), n,
                                 ptr<double>(h->td_theta), ptr<double>(h->td_work));
  SC_TRY(check_last(h, "dense eigenvalue launch"));
  h->spectrum.resize(n);
  SC_HIP(h, hipMemcpyAsync(h->spectrum.data(), h->td_theta.p, (size_t)n * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(h->h_flags + 12, ptr<int>(h->flags) + 12, sizeof(int),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(h->spectrum[i])) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
  return SC_OK;
}

// The leading `cols` eigenvectors of Op (largest eigenvalues first) from the tridiagonal form
// dense_spectrum left behind (reflectors in `scratch`, taus, d, e; h->spectrum): inverse
// iteration on T (host), Q z on the device, then the usual t .* u / ||.|| back-transform.
// The landing pad of every spectrum block Lanczos gives up on: like np.linalg.eig
// (utils.py:59) it always returns.
static int dense_vectors(sc_handle h, const double* scratch, int ld, int n, int cols) {
  hipStream_t s = h->stream;
  if (cols < 1 || cols > n)
    return fail(h, SC_ERR_UNSUPPORTED, "dense eigenvector request out of range");
  SC_TRY(ensure_vectors(h, n, cols));
  std::vector<double> de(2 * (size_t)n);
  SC_HIP(h, hipMemcpyAsync(de.data(), h->td_d.p, (size_t)n * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(de.data() + n, h->td_e.p, (size_t)n * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  const int lde = round_up(n, 16);
  std::vector<double> z((size_t)lde * cols, 0.0);
  if (!host_tridiag_eigvectors(de.data(), de.data() + n, n, h->spectrum.data(), cols, z.data(),
                               (size_t)lde))
    return fail(h, SC_ERR_NOT_CONVERGED, "inverse iteration on the tridiagonal form failed");
  SC_HIP(h, hipMemcpyAsync(h->E.p, z.data(), z.size() * sizeof(double), hipMemcpyHostToDevice,
                           s));
  launch_td_backtransform(s, scratch, ld, n, ptr<double>(h->td_tau), ptr<double>(h->E), lde,
                          cols);
  back_transform_cols(h, n, cols);
  SC_TRY(check_last(h, "dense eigenvector launch"));
  SC_HIP(h, hipStreamSynchronize(s));  // z is a local
  h->n_vec = cols;
  return SC_OK;
}

// Does the request read the whole spectrum (or its far end) exactly?
bool wants_full_spectrum(const EigRequest& rq) {
  if (rq.fixed_count > 0 || rq.descend) return false;
  // ascending: every eigenvalue when max_clusters is None (utils.py:100-115); the
  // NormalizedDiff gap divides by np.max(eigenvalues), the far end of the spectrum, which
  // no Krylov method resolves to 1e-5 on a dense bulk edge
  return rq.max_clusters == 0 || rq.eigengap_type == SC_EIGENGAP_NORMALIZED_DIFF;
}

static EigWorkspace eig_workspace(sc_handle h) {
  EigWorkspace ws;
  ws.Q = ptr<double>(h->Q);
  ws.Q2 = ptr<double>(h->Q2);
  ws.Vs = ptr<double>(h->Vs);
  ws.W = ptr<double>(h->W);
  ws.partial = ptr<double>(h->partial);
  ws.T = ptr<double>(h->T);
  ws.Y = ptr<double>(h->Y);
  ws.theta = ptr<double>(h->theta);
  ws.resid = ptr<double>(h->resid);
  ws.G = ptr<double>(h->G);
  ws.Rinv = ptr<double>(h->Rinv);
  ws.Hbuf = ptr<double>(h->Hbuf);
  ws.hsq = ptr<double>(h->hsq);
  ws.Yt = ptr<double>(h->Yt);
  ws.colnorm = ptr<double>(h->colnorm);
  ws.flags = ptr<int>(h->flags);
  return ws;
}

// S (n x n, ld) symmetric on the device; cvec/pvec/tvec already set.
// With h->free_on (matrix-free Diffuse, free_api.hip) `S_in` is the symmetric matrix A BEFORE
// Diffuse and the operator is diag(p) + diag(c) A A diag(c): two block products per pass,
// S = A A^T is only formed if a dense route needs its entries.
int sym_topk(sc_handle h, const double* S_in, int ld, int n, const EigRequest& rq_in,
             sc_diag* diag, EigDecision* out_dc, std::vector<double>* out_w,
             double* scratch_in) {
  hipStream_t s = h->stream;
  SC_TRY(ensure_eig(h, n));
  const double* S = S_in;
  double* scratch = scratch_in;
  const bool free_at_entry = h->free_on;
  // (set by eig_ncluster_impl when its scaling kernel has just cleared flags[13..15]; consumed
  //  here whatever route this solve takes, so that it never outlives the call it was set for)
  bool chain_flags_clean = h->chain_flags_clean;
  h->chain_flags_clean = false;
  // S = A A^T after all (a dense route reads entries; or the exact-row route gave up): the fp64
  // MFMA product into the scratch matrix, A's buffer becomes the scratch
  auto free_materialize = [&](bool with_stats) -> int {
    if (!h->free_on) return SC_OK;
    if (scratch == nullptr || scratch == S)
      return fail(h, SC_ERR_UNSUPPORTED, "no scratch matrix to form the Diffuse product in");
    SC_TRY(ensure_tilemap(h, n));
    GemmRowStats rs{1, ptr<double>(h->statp), ptr<double>(h->statp) + (size_t)n * gemm_tile_dim(n),
                    ptr<double>(h->rowmax), ptr<double>(h->rowsum)};
    launch_gemm_nt(s, S, ld, S, ld, scratch, ld, n, n, n, kEpiNone, true, ptr<double>(h->splitk),
                   h->tilemap_cur, with_stats ? &rs : nullptr);
    SC_TRY(check_last(h, "diffuse launch"));
    double* a_buffer = const_cast<double*>(S);
    S = scratch;
    scratch = a_buffer;
    h->free_on = false;
    return SC_OK;
  };
  // rows whose candidate list overflowed are evaluated in full once the stream has drained;
  // *restart: the scaling vectors changed under a solve that had already started
  auto free_check = [&](bool* restart) -> int {
    *restart = false;
    if (!h->free_on || h->free_checked) return SC_OK;
    bool changed = false, too_many = false;
    SC_TRY(free_fix_overflow(h, S, ld, n, &changed, &too_many));
    if (too_many) {
      SC_TRY(free_materialize(true));
      changed = true;
    }
    if (changed) {
      launch_scaling_vectors(s, ptr<double>(h->rowmax), ptr<double>(h->rowsum), n, h->free_lap,
                             h->free_rownorm, ptr<double>(h->cvec), ptr<double>(h->pvec),
                             ptr<double>(h->tvec));
      launch_check_finite(s, ptr<double>(h->cvec), ptr<double>(h->pvec), n, ptr<int>(h->flags) + 12);
      SC_TRY(check_last(h, "scaling launch"));
      *restart = true;
    }
    return SC_OK;
  };
  double* theta_d = ptr<double>(h->theta);
  double* resid_d = ptr<double>(h->resid);
  const double* cvec = ptr<double>(h->cvec);
  const double* pvec = ptr<double>(h->pvec);
  EigDecision dc;
  int m = 0, passes = 0, cycles = 0;
  EigRequest rq = rq_in;
  // (eig_skip_fused: the lockstep group solve saw this problem latch the fused chain)
  bool fused = !sw::eig_host_chain() && !h->eig_skip_fused;
  h->eig_skip_fused = false;
  bool three_pass = false;  // second attempt of the fused chain, see LzChain::three_pass
  // upper-triangle matvec once the matrix no longer fits the caches (below that the full
  // read is served on-die and the second launch costs more than it saves)
  const int sym_min_n = sw::matvec_sym_min_n();
  const bool sym_mv = n >= sym_min_n;
  // Dense full-spectrum route (n > 128): all eigenvalues from the tridiagonal form, the
  // eigengap decision from those, then the same Lanczos loop below for just the vectors.
  EigDecision dense_dc;
  bool dense = false, many_vectors = false;
  auto run_dense = [&]() -> int {
    bool unused = false;
    SC_TRY(free_check(&unused));
    SC_TRY(free_materialize(false));
    SC_TRY(dense_spectrum(h, S, ld, n, scratch));
    std::vector<double> zeros(n, 0.0);
    dense_dc = analyze(rq_in, h->spectrum.data(), zeros.data(), n, n, true);
    if (!dense_dc.enough) return fail(h, SC_ERR_UNSUPPORTED, "eigen request cannot be satisfied");
    // More vectors than a Krylov basis comfortably yields (the eigengap selected > 64
    // clusters, min_clusters > 64, or a stage request for > 64 pairs): they come from the
    // tridiagonal form too (dense_vectors), like every vector of the landing pad.
    many_vectors = dense_dc.kvec > kMaxVectors;
    dense = true;
    rq = rq_in;
    rq.fixed_count = std::max(1, dense_dc.kvec);  // vectors only
    rq.value_tol = std::max(rq_in.value_tol, 1e-6);
    return SC_OK;
  };
  if (n > kDenseMax && wants_full_spectrum(rq_in)) SC_TRY(run_dense());
  // Block Lanczos gave up (`reason`: 1 restart budget, 2 projected problem, 3 no full-rank
  // block): eigenvalues AND eigenvectors from the tridiagonal form.  np.linalg.eig
  // (utils.py:59) always returns, so must this.
  bool vectors_from_dense = false;
  int fallback_reason = 0;
  auto dense_fallback = [&](int reason) -> int {
    if (scratch == nullptr || scratch == S)
      return fail(h, SC_ERR_NOT_CONVERGED,
                  "block Lanczos did not converge and no scratch matrix is free for the dense path");
    if (sw::eig_trace())
      fprintf(stderr, "[sc] block Lanczos gave up (reason %d, %d passes): dense path\n", reason,
              passes);
    if (!dense) SC_TRY(run_dense());
    SC_TRY(free_materialize(false));  // (a `dense` that predates the hand-over: not in free mode)
    int cols = dense_dc.kvec;
    if (rq_in.fixed_count > 0 || rq_in.max_clusters > 0)
      cols = std::max(std::min(dense_dc.kw, kMaxVectors), cols);
    cols = std::max(1, cols);
    SC_TRY(dense_vectors(h, scratch, ld, n, cols));
    vectors_from_dense = true;
    fallback_reason = reason;
    return SC_OK;
  };

  if (n <= kDenseMax) {
    // ---- direct dense path: every eigenpair, one Jacobi launch
    launch_jacobi(s, S, ld, n, 1, cvec, pvec, nullptr, theta_d, ptr<double>(h->Y), kLdq,
                  nullptr, ptr<double>(h->Yt), ptr<int>(h->flags));
    SC_TRY(check_last(h, "jacobi launch"));
    SC_HIP(h, hipMemcpyAsync(h->h_theta, theta_d, n * sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipMemcpyAsync(h->h_flags + 12, ptr<int>(h->flags) + 12, sizeof(int),
                             hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipStreamSynchronize(s));
    if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
    for (int i = 0; i < n; ++i) h->h_theta[kLdq + i] = 0.0;
    dc = analyze(rq, h->h_theta, h->h_theta + kLdq, n, n, true);
    if (!dc.enough) return fail(h, SC_ERR_UNSUPPORTED, "eigen request cannot be satisfied");
    m = n;
    const int cols = n;  // all eigenvectors, like np.linalg.eig
    launch_rowmajor_to_colmajor(s, ptr<double>(h->Y), kLdq, n, cols, ptr<double>(h->E),
                                round_up(n, 16));
    back_transform_cols(h, n, cols);
    h->n_vec = cols;
    if (diag) diag->eig_path = SC_EIG_PATH_DENSE_JACOBI;
    dc.kw = n;
  } else {
  restart_lanczos:
    m = 0;
    cycles = 0;
    bool done = false;
    uint64_t seed = 0x5eed5eedull;
    const double* vscale = h->vs_scale ? h->vs_scale : cvec;
    LzChain chain;
    if (dense && many_vectors) {
      // the full spectrum is known and more than 64 vectors are wanted: all of them from the
      // tridiagonal form (inverse iteration + Householder back-transform), no Krylov solve
      SC_TRY(dense_fallback(5));
      done = true;
    }
    // ---- start block
    if (done) {
    } else if (fused) {
      // fused chain (k_lz_step): no host synchronisation until the first Rayleigh-Ritz
      // (the first start of a call: the scaling kernel has just cleared them, api.hip)
      if (!chain_flags_clean)
        SC_HIP(h, hipMemsetAsync(ptr<int>(h->flags) + 13, 0, 3 * sizeof(int), s));
      chain_flags_clean = false;
      const EigWorkspace ws = eig_workspace(h);
      chain = LzChain();
      chain.three_pass = three_pass;
      // random block + its Gram | CholQR | CholQR again (Gram checked against I) + store
      launch_lz_link(s, ws, &chain, n, 0, 0, 4, -1, vscale, 0, true, seed, false);
      launch_lz_link(s, ws, &chain, n, 0, 4, 3, -1, vscale, 0, false, 0, true);
      launch_lz_link(s, ws, &chain, n, 0, 3, 0, 0, vscale, 0, false, 0, false);
      SC_TRY(check_last(h, "start block launch"));
    } else {
      launch_random_block(s, ptr<double>(h->W), n, seed);
      SC_TRY(orthonormalize(h, n, 0, false, 0, 0, false));
      const int rc0 = finish_block(h, n, 0, 0, &seed);
      if (rc0 == SC_ERR_NOT_CONVERGED) {
        SC_TRY(dense_fallback(3));
        done = true;
      } else {
        SC_TRY(rc0);
      }
      SC_HIP(h, hipMemsetAsync(h->T.p, 0, (size_t)kLdq * kLdq * sizeof(double), s));
    }
    // test switch: take the landing pad straight away (tests/test_gpu_alternate_paths.py)
    if (!done && sw::eig_force_dense()) {
      SC_TRY(dense_fallback(4));
      done = true;
    }
    // basis cap: LDS Jacobi limit, and basis + next block must fit in R^n
    const int cap = std::min(kEigBasisCap, ((n - kEigBlock) / kEigBlock) * kEigBlock);
    // first Rayleigh-Ritz check after 3 blocks.  (Round 2 moved it to where the previous
    // solve on this handle had converged; that made the basis size at exit -- hence results
    // at the tolerance level -- depend on the call history.  A solve is now a function of
    // its input alone: tests/test_gpu_predict.py::test_results_do_not_depend_on_call_history.)
    const int first_check = std::min(3 * kEigBlock, cap);
    // Rayleigh-Ritz is the expensive serial step: every block early on (where convergence is
    // expected), then sparser, then once per restart cycle.
    auto check_due = [&](int mm) {
      return cycles == 0 ? (mm >= first_check && (mm <= 4 * kEigBlock ||
                                                  mm % (2 * kEigBlock) == 0 || mm + kEigBlock > cap))
                         : (mm + kEigBlock > cap);
    };
    // one block step from a basis of m_before vectors: the operator on the block
    // V_j = Q[:, m_before : m_before + 8] (Vs = c .* V_j), then the orthonormalisation chain
    auto enqueue_step = [&](int m_before) -> int {
      const bool time_mv = h->profile_level >= 2 && h->n_mv_ev < 16;
      if (time_mv) ev_rec(h, &h->mv_ev[h->n_mv_ev][0]);
      if (h->free_on)
        free_apply_operator(h, S, ld, n, sym_mv, ptr<double>(h->Q) + m_before, kLdq);
      else if (sym_mv)
        launch_block_matvec_sym(s, S, ld, n, cvec, pvec, ptr<double>(h->Q) + m_before, kLdq,
                                ptr<double>(h->Vs), ptr<double>(h->W), ptr<double>(h->mvsym));
      else
        launch_block_matvec(s, S, ld, n, cvec, pvec, ptr<double>(h->Q) + m_before, kLdq,
                            ptr<double>(h->Vs), ptr<double>(h->W));
      if (time_mv) ev_rec(h, &h->mv_ev[h->n_mv_ev++][1]);
      ++passes;
      const int mm = m_before + kEigBlock;
      if (fused) {
        // CGS-1 | CGS-2 + CholQR | re-projection + CholQR on the normalised block | store
        const EigWorkspace ws = eig_workspace(h);
        launch_lz_link(s, ws, &chain, n, mm, 0, 1, -1, vscale, mm - kEigBlock, false, 0, false);
        launch_lz_link(s, ws, &chain, n, mm, 1, 2, -1, vscale, mm - kEigBlock, false, 0, false);
        launch_lz_link(s, ws, &chain, n, mm, 2, 3, -1, vscale, mm - kEigBlock, false, 0, false);
        if (three_pass) launch_lz_link(s, ws, &chain, n, mm, 3, 3, -1, vscale, 0, false, 0, false);
        launch_lz_link(s, ws, &chain, n, mm, 3, 0, mm, vscale, 0, false, 0, false);
        return check_last(h, "block step launch");
      }
      return orthonormalize(h, n, mm, true, mm - kEigBlock, mm, true);
    };
    int ahead = 0;  // block steps already enqueued beyond m (run-ahead during a host solve)
    while (!done) {
      if (ahead > 0)
        --ahead;
      else
        SC_TRY(enqueue_step(m));
      m += kEigBlock;
      const bool check = check_due(m);
      const bool host_rr = check && m <= kHostRRSingle && !sw::eig_device_rr();
      if (host_rr) {
        // small projected problem: T and G come back with the flags; solved on the host
        SC_HIP(h, hipMemcpy2DAsync(h->h_rr, (size_t)m * sizeof(double), h->T.p,
                                   (size_t)kLdq * sizeof(double), (size_t)m * sizeof(double), m,
                                   hipMemcpyDeviceToHost, s));
        SC_HIP(h, hipMemcpyAsync(h->h_rr + kHostRRSingle * kHostRRSingle, h->G.p,
                                 kEigBlock * kEigBlock * sizeof(double), hipMemcpyDeviceToHost,
                                 s));
      } else if (check) {
        launch_jacobi(s, ptr<double>(h->T), kLdq, m, 0, nullptr, nullptr, ptr<double>(h->G),
                      theta_d, ptr<double>(h->Y), kLdq, resid_d, ptr<double>(h->Yt),
                      ptr<int>(h->flags));
        SC_TRY(check_last(h, "jacobi launch"));
        SC_HIP(h, hipMemcpyAsync(h->h_theta, theta_d, m * sizeof(double),
                                 hipMemcpyDeviceToHost, s));
        SC_HIP(h, hipMemcpyAsync(h->h_theta + kLdq, resid_d, m * sizeof(double),
                                 hipMemcpyDeviceToHost, s));
      }
      if (!fused) {
        const int rcb = finish_block(h, n, m, m, &seed);  // syncs the stream
        if (rcb == SC_ERR_NOT_CONVERGED) {
          SC_TRY(dense_fallback(3));
          break;
        }
        SC_TRY(rcb);
      } else if (check) {
        // the one synchronisation of the fused chain.  From the third check of the first
        // cycle on (a spectrum that did not converge in 32 vectors will not in 48 either,
        // more often than not) the device runs ahead to the next check point while the host
        // waits for and solves this one: the projected problems of 48-128 vectors take
        // 0.13-0.93 ms on the host, a block step 0.1-0.25 ms on the device.  What the check
        // reads -- T[0:m, 0:m], G, the flags -- is copied out before the run-ahead in stream
        // order; what a converged check uses -- Q[:, 0:m] -- is not touched by it.
        SC_HIP(h, hipMemcpyAsync(h->h_flags, h->flags.p, 16 * sizeof(int), hipMemcpyDeviceToHost, s));
        if (!h->sync_ev) SC_HIP(h, hipEventCreateWithFlags(&h->sync_ev, hipEventDisableTiming));
        SC_HIP(h, hipEventRecord(h->sync_ev, s));
        if (host_rr && cycles == 0 && m >= 6 * kEigBlock && m + kEigBlock <= cap) {
          int mm = m;
          do {
            SC_TRY(enqueue_step(mm));
            mm += kEigBlock;
            ++ahead;
          } while (!check_due(mm));
        }
        SC_HIP(h, hipEventSynchronize(h->sync_ev));
        if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
        if (h->h_flags[13] != 0) {
          // a dependent column or a hopeless first Cholesky somewhere in the chain: redo the
          // solve with the host-driven chain, which repairs blocks one by one
          passes = 0;
          if (!three_pass && h->h_flags[15] == 2000) {
            // only "Gram far from I after the first pass" (an ill-conditioned block, no
            // dependent column): once more with the three-pass chain
            if (sw::eig_trace())
              fprintf(stderr, "[sc] fused chain flagged at m=%d: three-pass chain\n",
                      h->h_flags[14]);
            three_pass = true;
            goto restart_lanczos;
          }
          if (sw::eig_trace()) fprintf(stderr, "[sc] fused chain flagged: host chain\n");
          fused = false;
          goto restart_lanczos;
        }
      }
      if (h->free_on && !h->free_checked && (!fused || check)) {
        // (the stream has just drained) matrix-free Diffuse: rows the candidate search could
        // not prune get their exact maximum now; the solve starts over on the corrected operator
        bool restart = false;
        SC_TRY(free_check(&restart));
        if (restart) {
          if (sw::eig_trace())
            fprintf(stderr, "[sc] matrix-free diffuse: %d rows evaluated in full, restart\n",
                    h->h_free[0]);
          passes = 0;
          goto restart_lanczos;
        }
      }
      if (host_rr) {
        double* hy = h->h_rr + kHostRRSingle * kHostRRSingle + 64;
        const double* hG = h->h_rr + kHostRRSingle * kHostRRSingle;
        bool rr_ok;
        timespec ts0;
        if (sw::eig_trace()) clock_gettime(CLOCK_MONOTONIC, &ts0);
        if (m <= kHostRR) {
          rr_ok = host_rayleigh_ritz(h->h_rr, m, hG, m, h->h_theta, h->h_theta + kLdq, hy, m);
        } else {
          // the pairs the analysis reads and a thick restart keeps (its `keep` below), + a block
          auto leading = [&](const double* theta) {
            std::vector<double> zero(m, 0.0);
            const EigDecision d0 = analyze(rq, theta, zero.data(), m, n, false);
            const int want = d0.enough ? std::max(d0.kw, d0.kvec) : cap / 4;
            return round_up(want + kEigBlock, kEigBlock) + kEigBlock;
          };
          rr_ok = host_rayleigh_ritz_leading(h->h_rr, m, hG, m, h->h_theta, h->h_theta + kLdq,
                                             hy, m, leading);
        }
        if (sw::eig_trace()) {
          timespec ts1;
          clock_gettime(CLOCK_MONOTONIC, &ts1);
          fprintf(stderr, "[sc]   host Rayleigh-Ritz m=%d: %.0f us\n", m,
                  (ts1.tv_sec - ts0.tv_sec) * 1e6 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-3);
        }
        if (!rr_ok) {
          SC_TRY(dense_fallback(2));
          break;
        }
        for (int i = 0; i < m; ++i)
          if (!std::isfinite(h->h_theta[i])) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
        // the Ritz vectors (and values, for a restart) go back to where k_jacobi leaves them
        SC_HIP(h, hipMemcpy2DAsync(h->Y.p, (size_t)kLdq * sizeof(double), hy,
                                   (size_t)m * sizeof(double), (size_t)m * sizeof(double), m,
                                   hipMemcpyHostToDevice, s));
        SC_HIP(h, hipMemcpyAsync(theta_d, h->h_theta, m * sizeof(double),
                                 hipMemcpyHostToDevice, s));
      }
      if (check) {
        dc = analyze(rq, h->h_theta, h->h_theta + kLdq, m, n, false);
        if (sw::eig_trace()) {
          int worst = 0;
          double wr = 0.0;
          for (int i = 0; i < std::min(m, dc.kw > 0 ? dc.kw : m); ++i) {
            const double r = h->h_theta[kLdq + i] / std::max(std::fabs(h->h_theta[i]), 1e-300);
            if (r > wr) { wr = r; worst = i; }
          }
          fprintf(stderr, "[sc] lanczos pass %d m=%d cycle %d: enough=%d conv=%d kw=%d kvec=%d "
                  "fail kind %d at %d (theta %.6g resid %.2e); far end theta=%.6g resid=%.2e\n",
                  passes, m, cycles, dc.enough, dc.converged, dc.kw, dc.kvec, dc.fail_kind,
                  dc.fail_index, dc.fail_index >= 0 ? h->h_theta[dc.fail_index] : 0.0,
                  dc.fail_index >= 0 ? h->h_theta[kLdq + dc.fail_index] : 0.0, h->h_theta[m - 1],
                  h->h_theta[kLdq + m - 1]);
          (void)wr; (void)worst;
        }
        if (dc.unsupported) {
          // more than 64 eigenvalues are read (max_clusters=None with a slowly decaying
          // spectrum): take all of them from the dense path, then come back for the vectors
          if (dense) return fail(h, SC_ERR_UNSUPPORTED, "eigen request cannot be satisfied");
          SC_TRY(run_dense());
          goto restart_lanczos;
        }
        if (dc.enough && dc.converged) {
          if (!dense && m < n && block_multiplicity_suspect(rq, dc, h->h_theta, m)) {
            // kEigBlock equal Ritz values in front of what the caller reads: an eigenvalue of
            // higher multiplicity than the block can show -- the dense path counts them
            if (sw::eig_trace())
              fprintf(stderr, "[sc] %d equal Ritz values ahead of the decisive gap: dense path\n",
                      kEigBlock);
            SC_TRY(dense_fallback(7));
            break;
          }
          done = true;
          break;
        }
      }
      if (m + kEigBlock > cap) {
        // ---- thick restart: keep the leading Ritz vectors + the new block
        if (++cycles > rq.max_cycles) {
          SC_TRY(dense_fallback(1));
          break;
        }
        int want = dc.enough ? std::max(dc.kw, dc.kvec) : cap / 4;
        int keep = round_up(want + kEigBlock, kEigBlock);
        keep = std::max(kEigBlock, std::min(keep, cap - 2 * kEigBlock));
        launch_basis_times_Y(s, ptr<double>(h->Q), kLdq, m, ptr<double>(h->Y), kLdq, keep,
                             ptr<double>(h->Q2), kLdq, n, 0);
        launch_copy_block(s, ptr<double>(h->Q) + m, kLdq, ptr<double>(h->Q2) + keep, kLdq, n,
                          kEigBlock);
        std::swap(h->Q, h->Q2);
        launch_set_diag_T(s, ptr<double>(h->T), kLdq, kLdq, theta_d, keep);
        SC_TRY(check_last(h, "restart launch"));
        m = keep;
      }
    }
    // (block steps enqueued ahead of a check that ended the solve were not part of it: the
    //  reported pass count is what the result was computed from.  They wrote Vs, W and basis
    //  columns beyond m only -- every path that goes on from here rebuilds those.)
    if (ahead > 0) passes -= ahead;
    if (vectors_from_dense) {
      dc.max_resid = 0.0;
      if (diag) diag->eig_path = SC_EIG_PATH_DENSE_FULL;
    } else {
      const int cols = std::min(std::max(dc.kw, dc.kvec), kMaxVectors);
      launch_basis_times_Y(s, ptr<double>(h->Q), kLdq, m, ptr<double>(h->Y), kLdq, cols,
                           ptr<double>(h->E), round_up(n, 16), n, 1);
      back_transform_cols(h, n, cols);
      SC_TRY(check_last(h, "ritz vector launch"));
      h->n_vec = cols;
      if (diag) diag->eig_path = dense ? SC_EIG_PATH_DENSE_TRIDIAG : SC_EIG_PATH_BLOCK_LANCZOS;
    }
  }
  if (dense) {
    // values and the eigengap decision come from the full spectrum; the Lanczos pass above
    // only supplied the vectors
    const double vec_resid = dc.max_resid;
    dc = dense_dc;
    dc.converged = true;
    dc.max_resid = vec_resid;
    if (out_w) {  // the whole spectrum is known: report all of it (np.max included)
      out_w->resize(n);
      for (int i = 0; i < n; ++i) (*out_w)[i] = rq.descend ? h->spectrum[i] : -h->spectrum[i];
    }
  } else if (out_w) {
    out_w->resize(dc.kw);
    for (int i = 0; i < dc.kw; ++i) (*out_w)[i] = rq.descend ? h->h_theta[i] : -h->h_theta[i];
  }
  if (diag) {
    diag->eig_matvec_passes = passes;
    diag->eig_block = kEigBlock;
    diag->eig_basis = m;
    diag->eig_cycles = cycles;
    diag->eig_max_residual = dc.max_resid;
    diag->eig_host_chain = (n > kDenseMax && !fused) ? 1 : 0;
    diag->eig_fallback = fallback_reason;
    if (free_at_entry) {
      diag->diffuse_path = h->free_on ? SC_DIFFUSE_PATH_FREE : SC_DIFFUSE_PATH_FREE_THEN_EXPLICIT;
      diag->free_candidates = h->free_checked ? h->h_free[65] : 0;
      diag->free_overflow_rows = h->free_checked ? h->h_free[0] : 0;
      diag->free_tiles_run = h->free_checked ? h->h_free[67] : 0;
    }
  }
  *out_dc = dc;
  return SC_OK;
}

// ------------------------------------------------------------------------------
// lockstep group solve (batch_group.hip)
// ------------------------------------------------------------------------------
// The common case of sym_topk -- fused chain, Rayleigh-Ritz on the host, convergence within
// kHostRR basis vectors, no restart -- for several independent problems at once.  Short
// utterances leave most of the chip idle and their solve is a chain of ~25 tiny dependent
// launches and two host synchronisations; here every launch carries one link (or one
// matvec) of EVERY member and one synchronisation serves all of their checks.  Each member's
// arithmetic is exactly sym_topk's (same kernels bodies, same arguments, a check after
// every block from the third on); whatever leaves the common case (a latched chain flag,
// non-finite input, no convergence by kHostRR vectors, a request for more values than a
// Krylov basis holds) is handed back with status 1 and goes through sym_topk itself.
static double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

bool sym_group_eligible(int n, const EigRequest& rq, bool any_size) {
  const int sym_min_n = sw::matvec_sym_min_n();
  return n > kDenseMax && (any_size || n < sym_min_n) && !wants_full_spectrum(rq) &&
         !sw::eig_host_chain() && !sw::eig_device_rr();
}

// pinned host + device staging of the group checks: per member m*m (T) + 64 (G) doubles +
// 16 flag words going out, kHostRR^2 (Y) doubles coming back
static int ensure_group_staging(sc_handle lead) {
  const size_t out_doubles = (size_t)kGroupMax * (kEigBasisCap * kEigBasisCap + 64 + 8);
  const size_t y_doubles = (size_t)kGroupMax * kEigBasisCap * kEigBasisCap;
  SC_TRY(grow(lead, lead->gpack, out_doubles * sizeof(double)));
  SC_TRY(grow(lead, lead->gypack, y_doubles * sizeof(double)));
  if (!lead->h_gpack) {
    SC_HIP(lead, hipHostMalloc(reinterpret_cast<void**>(&lead->h_gpack),
                               out_doubles * sizeof(double)));
    SC_HIP(lead, hipHostMalloc(reinterpret_cast<void**>(&lead->h_gypack),
                               y_doubles * sizeof(double)));
    SC_HIP(lead, hipEventCreateWithFlags(&lead->gcheck_ev, hipEventDisableTiming));
  }
  if (!lead->gpool) {
    const unsigned hw = std::thread::hardware_concurrency();
    // one worker per member of a full group where the host has the cores for it (a sweep of
    // slowly converging values spends half its time in these solves: 16 members on 7 threads
    // are three rounds of up to 0.9 ms per check); three lanes of a batch have a pool each
    lead->gpool = new HostPool(hw >= 64 ? kGroupMax - 1 : (hw >= 8 ? 6 : (hw >= 4 ? 2 : 0)));
  }
  return SC_OK;
}

int sym_topk_group(sc_handle lead, GroupEigMember* mem, int count, bool want_vectors) {
  if (count < 1 || count > kGroupMax) return fail(lead, SC_ERR_INVALID, "group size");
  hipStream_t s = lead->stream;
  SC_TRY(ensure_group_staging(lead));
  LzGroupMember lz[kGroupMax];
  int limit[kGroupMax];
  for (int z = 0; z < count; ++z) {
    sc_handle h = mem[z].h;
    SC_TRY(ensure_eig(h, mem[z].n));  // (flags[13..15] were cleared with the member's front)
    lz[z].ws = eig_workspace(h);
    lz[z].chain = LzChain();
    lz[z].chain.three_pass = true;  // (one more tiny launch per block for the whole group)
    lz[z].n = mem[z].n;
    lz[z].vs_scale = h->vs_scale ? h->vs_scale : ptr<double>(h->cvec);
    lz[z].active = true;
    const int cap = std::min(kEigBasisCap, ((mem[z].n - kEigBlock) / kEigBlock) * kEigBlock);
    limit[z] = cap;
    mem[z].status = 0;
    mem[z].passes = 0;
    mem[z].basis = 0;
  }
  // upper-triangle matvec once the group's matrices no longer fit the caches together
  size_t matrix_bytes = 0;
  for (int z = 0; z < count; ++z) matrix_bytes += (size_t)mem[z].n * mem[z].ld * sizeof(double);
  const size_t sym_min_bytes = (size_t)128 << 20;
  const bool sym_matvec = matrix_bytes >= sym_min_bytes;
  bool any_free = false;
  for (int z = 0; z < count; ++z) any_free = any_free || mem[z].free_op;
  const uint64_t seed = 0x5eed5eedull;
  launch_lz_link_group(s, lz, count, 0, 0, 4, -1, 0, true, seed, false);
  launch_lz_link_group(s, lz, count, 0, 4, 3, -1, 0, false, 0, true);
  launch_lz_link_group(s, lz, count, 0, 3, 0, 0, 0, false, 0, false);
  SC_TRY(check_last(lead, "group start block launch"));
  // first Rayleigh-Ritz check after three blocks -- after four once the batch so far says that
  // (almost) nobody is done by three (a check is a host solve per member)
  int first_check = 3 * kEigBlock;
  if (lead->gconv_seen >= 2 * kGroupMax && lead->gconv_hist[3] * 20 < lead->gconv_seen)
    first_check = 4 * kEigBlock;
  int active = count;
  const bool trace = sw::group_trace();
  double us_first_sync = 0.0, us_sync = 0.0, us_host = 0.0;
  int steps = 0, wasted = 0;
  // one block step of every active member: matvec + the five links of the three-pass chain
  auto block_step = [&](int m_before) -> int {
    MatvecItem mv[kGroupMax];
    memset(mv, 0, sizeof(mv));
    for (int z = 0; z < count; ++z) {
      if (!lz[z].active) continue;
      sc_handle h = mem[z].h;
      mv[z].S = mem[z].S;
      mv[z].ld = mem[z].ld;
      mv[z].n = mem[z].n;
      mv[z].cvec = ptr<double>(h->cvec);
      mv[z].pvec = ptr<double>(h->pvec);
      mv[z].V = ptr<double>(h->Q) + m_before;
      mv[z].ldv = kLdq;
      mv[z].Vs = ptr<double>(h->Vs);
      mv[z].W = ptr<double>(h->W);
      mv[z].slabs = ptr<double>(h->mvsym);
      ++mem[z].passes;
    }
    if (any_free) {
      // matrix-free members: fY = A Vs first (c = 1, p = 0), then W = p .* V + c .* (A fY)
      MatvecItem inner[kGroupMax];
      memset(inner, 0, sizeof(inner));
      for (int z = 0; z < count; ++z) {
        if (!lz[z].active || !mem[z].free_op) continue;
        inner[z] = mv[z];
        inner[z].cvec = inner[z].pvec = nullptr;
        inner[z].V = mv[z].Vs;
        inner[z].ldv = kEigBlock;
        inner[z].W = ptr<double>(mem[z].h->fY);
        mv[z].Vs = ptr<double>(mem[z].h->fY);
      }
      launch_block_matvec_group(s, inner, count, sym_matvec);
    }
    launch_block_matvec_group(s, mv, count, sym_matvec);
    const int m = m_before + kEigBlock;
    launch_lz_link_group(s, lz, count, m, 0, 1, -1, m - kEigBlock, false, 0, false);
    launch_lz_link_group(s, lz, count, m, 1, 2, -1, m - kEigBlock, false, 0, false);
    launch_lz_link_group(s, lz, count, m, 2, 3, -1, m - kEigBlock, false, 0, false);
    launch_lz_link_group(s, lz, count, m, 3, 3, -1, 0, false, 0, false);
    launch_lz_link_group(s, lz, count, m, 3, 0, m, 0, false, 0, false);
    ++steps;
    return check_last(lead, "group block step launch");
  };
  // T (m x m), the residual Gram and the flags of every active member: one gather kernel, one
  // copy to the host, an event to wait on
  const int out_stride = kEigBasisCap * kEigBasisCap + 64 + 8;
  const int y_stride = kEigBasisCap * kEigBasisCap;
  // Round 4: the group no longer ends at kHostRR basis vectors.  Members that have not
  // converged keep growing their bases in lockstep up to the cap (the leading-vector
  // Rayleigh-Ritz of sym_topk above 64 vectors), and at the cap the whole group makes a thick
  // restart -- same kept size for every member, so that they stay in lockstep -- instead of
  // each member being solved again from scratch on the single-call path (binarised
  // affinities, unstructured embeddings: 28-64 passes per solve).
  int group_cap = kEigBasisCap;
  for (int z = 0; z < count; ++z) group_cap = std::min(group_cap, limit[z]);
  int cycles = 0, max_cycles = 1 << 30;
  for (int z = 0; z < count; ++z) max_cycles = std::min(max_cycles, mem[z].rq.max_cycles);
  auto request_check = [&](int m) -> int {
    GatherItem gi[kGroupMax];
    memset(gi, 0, sizeof(gi));
    for (int z = 0; z < count; ++z) {
      if (!lz[z].active) continue;
      sc_handle h = mem[z].h;
      gi[z].T = ptr<double>(h->T);
      gi[z].G = ptr<double>(h->G);
      gi[z].flags = ptr<int>(h->flags);
    }
    launch_group_gather(s, gi, count, m, ptr<double>(lead->gpack), out_stride);
    SC_HIP(lead, hipMemcpyAsync(lead->h_gpack, lead->gpack.p,
                                (size_t)count * out_stride * sizeof(double),
                                hipMemcpyDeviceToHost, s));
    SC_HIP(lead, hipEventRecord(lead->gcheck_ev, s));
    return SC_OK;
  };
  int m = 0;
  while (m < first_check) {
    SC_TRY(block_step(m));
    m += kEigBlock;
  }
  SC_TRY(request_check(m));
  bool any_solved = false;
  while (active > 0) {
    // While the host solves the projected problems of this check the device takes every
    // active member one block further.  For a member that turns out converged (or is handed
    // back) that block is wasted but harmless: what its results are made of -- Q[:, 0:m],
    // the Ritz coefficients -- is not touched by it.  Skipped where the batch so far says
    // that nearly every member is done by this size.
    const int seen = lead->gconv_seen;
    int done_by = 0;
    for (int b = 0; b <= m / kEigBlock && b < 16; ++b) done_by += lead->gconv_hist[b];
    // (never speculating costs 3 % on config 5; speculating while most members are already
    //  done wastes a matvec pass over their matrices: stop once half of the members seen so
    //  far in this batch had converged by this basis size)
    // (never at the basis cap: the next block has no room, a restart follows)
    const bool speculate = (seen < 2 * kGroupMax || done_by * 2 < seen) && m + kEigBlock <= group_cap;
    if (speculate) SC_TRY(block_step(m));
    const double t_sync0 = trace ? now_us() : 0.0;
    SC_HIP(lead, hipEventSynchronize(lead->gcheck_ev));
    const double t_sync1 = trace ? now_us() : 0.0;
    if (trace) (m == first_check ? us_first_sync : us_sync) += t_sync1 - t_sync0;
    const int active_before = active;
    // The projected problems of the members are independent: solved on the lead handle's few
    // host workers + this thread (one after the other they were 1.0-1.9 ms per group, during
    // which this stream has nothing queued but the speculative block).
    bool rr_wanted[kGroupMax], rr_ok[kGroupMax];
    int keep_want[kGroupMax] = {0};
    for (int z = 0; z < count; ++z) {
      rr_wanted[z] = rr_ok[z] = false;
      if (!lz[z].active) continue;
      const double* pack = lead->h_gpack + (size_t)z * out_stride;
      const int* hflags = reinterpret_cast<const int*>(pack + m * m + 64);
      const bool latched = hflags[13] != 0;
      rr_wanted[z] = !(hflags[12] != 0 || (latched && hflags[14] != m));
    }
    lead->gpool->run(count, [&](int z) {
      if (!rr_wanted[z]) return;
      sc_handle h = mem[z].h;
      const double* pack = lead->h_gpack + (size_t)z * out_stride;
      double* hy = lead->h_gypack + (size_t)z * y_stride;
      bool ok;
      if (m <= kHostRR) {
        ok = host_rayleigh_ritz(pack, m, pack + m * m, m, h->h_theta, h->h_theta + kLdq, hy, m);
      } else {
        // the pairs the analysis reads and a thick restart keeps, + a block (as sym_topk)
        const EigRequest& rqz = mem[z].rq;
        const int nz = mem[z].n;
        auto leading = [&](const double* theta) {
          // at the cap a restart follows whose kept size is the GROUP's: every vector it
          // could keep
          if (m + kEigBlock > group_cap) return m - kEigBlock;
          std::vector<double> zero(m, 0.0);
          const EigDecision d0 = analyze(rqz, theta, zero.data(), m, nz, false);
          const int want = d0.enough ? std::max(d0.kw, d0.kvec) : group_cap / 4;
          return round_up(want + kEigBlock, kEigBlock) + kEigBlock;
        };
        ok = host_rayleigh_ritz_leading(pack, m, pack + m * m, m, h->h_theta, h->h_theta + kLdq, hy,
                                        m, leading);
      }
      for (int i = 0; ok && i < m; ++i) ok = std::isfinite(h->h_theta[i]);
      rr_ok[z] = ok;
    });
    for (int z = 0; z < count; ++z) {
      if (!lz[z].active) continue;
      sc_handle h = mem[z].h;
      const double* pack = lead->h_gpack + (size_t)z * out_stride;
      const int* hflags = reinterpret_cast<const int*>(pack + m * m + 64);
      auto hand_back = [&]() {
        mem[z].status = 1;
        lz[z].active = false;
        --active;
      };
      // A latched chain (a Krylov block that is linearly dependent at working precision: the
      // basis spans an invariant subspace, the usual end of a numerically low-rank operator)
      // cannot go on, but if it latched in THIS block everything the check reads is intact
      // -- and the Ritz pairs of an invariant subspace are converged.
      const bool latched = hflags[13] != 0;
      if (trace && latched)
        fprintf(stderr, "[sc]   member %d (n %d): chain latched at m=%d (code %d), check at m=%d\n",
                z, mem[z].n, hflags[14], hflags[15], m);
      if (!rr_wanted[z]) {
        if (latched) h->eig_skip_fused = true;
        hand_back();
        continue;
      }
      if (!rr_ok[z]) {
        hand_back();
        continue;
      }
      const EigDecision dc = analyze(mem[z].rq, h->h_theta, h->h_theta + kLdq, m, mem[z].n, false);
      if (dc.unsupported) {
        hand_back();
        continue;
      }
      if (dc.enough && dc.converged && m < mem[z].n &&
          block_multiplicity_suspect(mem[z].rq, dc, h->h_theta, m)) {
        hand_back();  // (sym_topk sees the same signature and takes the dense path)
        continue;
      }
      if (dc.enough && dc.converged) {
        mem[z].dc = dc;
        mem[z].basis = m;  // its Ritz coefficients stay in h_gypack[z] until the end
        mem[z].w.resize(dc.kw);
        for (int i = 0; i < dc.kw; ++i)
          mem[z].w[i] = mem[z].rq.descend ? h->h_theta[i] : -h->h_theta[i];
        lz[z].active = false;
        --active;
        any_solved = true;
        if (cycles == 0 && m / kEigBlock < 16) ++lead->gconv_hist[m / kEigBlock];
        ++lead->gconv_seen;
        continue;
      }
      if (latched) {
        if (trace) fprintf(stderr, "[sc]   member %d: latched and not converged\n", z);
        h->eig_skip_fused = true;
        hand_back();
        continue;
      }
      keep_want[z] = dc.enough ? std::max(dc.kw, dc.kvec) : group_cap / 4;
    }
    if (trace) us_host += now_us() - t_sync1;
    if (speculate) wasted += active_before - active;
    if (active == 0) break;
    bool speculated = speculate;
    if (m + kEigBlock > group_cap) {
      // ---- thick restart of every member still active: [leading Ritz vectors | current block]
      if (++cycles > max_cycles) {  // restart budget spent: the single-call path lands them
        for (int z = 0; z < count; ++z)
          if (lz[z].active) {
            mem[z].status = 1;
            lz[z].active = false;
            --active;
          }
        break;
      }
      int keep = kEigBlock;
      for (int z = 0; z < count; ++z)
        if (lz[z].active) keep = std::max(keep, round_up(keep_want[z] + kEigBlock, kEigBlock));
      keep = std::max(kEigBlock, std::min(keep, group_cap - 2 * kEigBlock));
      for (int z = 0; z < count; ++z) {
        if (!lz[z].active) continue;
        sc_handle h = mem[z].h;
        const double* hy = lead->h_gypack + (size_t)z * y_stride;
        // Ritz coefficients and values of this member -> its arena (where k_jacobi leaves them)
        SC_HIP(lead, hipMemcpy2DAsync(h->Y.p, (size_t)kLdq * sizeof(double), hy,
                                      (size_t)m * sizeof(double), (size_t)m * sizeof(double), m,
                                      hipMemcpyHostToDevice, s));
        SC_HIP(lead, hipMemcpyAsync(h->theta.p, h->h_theta, m * sizeof(double),
                                    hipMemcpyHostToDevice, s));
        // (the speculative block step, if one ran, wrote Q[:, m + 8 ..] and W: the block that
        //  continues the recurrence is Q[:, m : m + 8], which the restart keeps)
        launch_basis_times_Y(s, ptr<double>(h->Q), kLdq, m, ptr<double>(h->Y), kLdq, keep,
                             ptr<double>(h->Q2), kLdq, mem[z].n, 0);
        launch_copy_block(s, ptr<double>(h->Q) + m, kLdq, ptr<double>(h->Q2) + keep, kLdq,
                          mem[z].n, kEigBlock);
        std::swap(h->Q, h->Q2);
        launch_set_diag_T(s, ptr<double>(h->T), kLdq, kLdq, ptr<double>(h->theta), keep);
        lz[z].ws = eig_workspace(h);
      }
      SC_TRY(check_last(lead, "group restart launch"));
      SC_HIP(lead, hipStreamSynchronize(s));  // (h_theta / h_gypack are rewritten by the next check)
      if (trace)
        fprintf(stderr, "[sc] group restart %d: %d members keep %d vectors\n", cycles, active, keep);
      m = keep;  // the recurrence continues from the kept block: Vs = c .* Q[:, m : m + 8] still holds
      speculated = false;
    }
    // next check: every block early in the first cycle (where convergence is expected), then
    // every other block, and only with a full basis once restarts have begun (sym_topk's rule)
    do {
      if (!speculated) SC_TRY(block_step(m));
      speculated = false;
      m += kEigBlock;
    } while (!(cycles == 0 ? (m <= 4 * kEigBlock || m % (2 * kEigBlock) == 0 || m + kEigBlock > group_cap)
                           : (m + kEigBlock > group_cap)));
    SC_TRY(request_check(m));
  }
  if (trace)
    fprintf(stderr, "[sc] group eigen: %d members, %d block steps (%d member-steps beyond "
            "convergence); wait at first check %.0f us, at later checks %.0f us, host "
            "Rayleigh-Ritz + analysis %.0f us\n", count, steps, wasted, us_first_sync, us_sync,
            us_host);
  // ---- Ritz vectors of every solved member: E = normalise(t .* (Q Y)), the coefficient
  //      matrices of all of them in one upload
  if (!any_solved || !want_vectors) return SC_OK;
  RitzItem rz[kGroupMax];
  memset(rz, 0, sizeof(rz));
  int last = -1;
  for (int z = 0; z < count; ++z) {
    if (mem[z].status != 0) continue;
    sc_handle h = mem[z].h;
    const int cols = std::min(std::max(mem[z].dc.kw, mem[z].dc.kvec), kMaxVectors);
    rz[z].Q = ptr<double>(h->Q);
    rz[z].ldq = kLdq;
    rz[z].m = mem[z].basis;
    rz[z].Y = ptr<double>(lead->gypack) + (size_t)z * y_stride;
    rz[z].ldy = mem[z].basis;
    rz[z].cols = cols;
    rz[z].E = ptr<double>(h->E);
    rz[z].lde = round_up(mem[z].n, 16);
    rz[z].n = mem[z].n;
    rz[z].tvec = ptr<double>(h->tvec);
    h->n_vec = cols;
    h->last_w = mem[z].w;
    last = z;
  }
  SC_HIP(lead, hipMemcpyAsync(lead->gypack.p, lead->h_gypack,
                              (size_t)(last + 1) * y_stride * sizeof(double),
                              hipMemcpyHostToDevice, s));
  launch_ritz_vectors_group(s, rz, count);
  return check_last(lead, "group ritz vector launch");
}

// ------------------------------------------------------------------------------
// general (non-symmetric) top-k eigensolver driver
// ------------------------------------------------------------------------------
// M (n x n, ld): the refined matrix, NOT diagonally similar to a symmetric one.
// Operator  Op x = p .* x + cl .* (M (cr .* x))  (= M, or minus the Laplacian), whose
// eigenvalues of largest real part are wanted; eigenvectors are those of the reference's
// matrix itself (no similarity transform).  n <= 64: the dense solver on the materialised
// matrix (every eigenpair, like np.linalg.eig).  Larger n: block Arnoldi with full
// re-orthogonalisation, explicit Rayleigh-Ritz H = Q^T Op Q (basis <= 64), explicit
// residuals ||Op v - theta v||, explicit restart from the wanted Ritz vectors (real and
// imaginary parts of complex pairs).

// Dense route for n > 64 (eig_path 7): every eigenvalue of the reference's own matrix -- M, or
// the Laplacian of M -- and the eigenvectors k-means reads.  Taken when the request reads more
// eigenvalues than a block Arnoldi basis holds (max_clusters=None with a Laplacian: all n of
// them, utils.py:100-115; max_clusters > 63; min_clusters > 64) and as the landing pad of a
// block Arnoldi that gives up: np.linalg.eig (utils.py:59) always returns.
//   device  the matrix is formed in `scratch` and reduced to Hessenberg form (hessenberg.hip)
//   host    QR iteration for the n eigenvalues, inverse iteration + back-transform for the
//           leading max(n_clusters, min_clusters) eigenvectors (host_eig.cpp)
//   device  dgeev's norm / phase convention and the real parts (k_gen_phase)
constexpr int kGenDenseLimit = 16384;  // (the host QR iteration is ~10 n^3 flops: minutes beyond)
static int gen_dense_large(sc_handle h, const double* M, int ld, int n, int laplacian_type,
                           const EigRequest& rq, sc_diag* diag, EigDecision* out_dc,
                           std::vector<double>* out_w, double* scratch, int reason) {
  hipStream_t s = h->stream;
  if (scratch == nullptr || scratch == M)
    return fail(h, SC_ERR_UNSUPPORTED, "no scratch matrix for the dense general eigen path");
  if (n > kGenDenseLimit)
    return fail(h, SC_ERR_UNSUPPORTED,
                "the dense general eigen path (every eigenvalue of a matrix that is not "
                "diagonally similar to a symmetric one) is limited to n <= 16384");
  const bool is_lap = laplacian_type >= SC_LAPLACIAN_UNNORMALIZED;
  if (sw::eig_trace())
    fprintf(stderr, "[sc] general eigen path: dense Hessenberg route, n=%d (reason %d)\n", n, reason);
  SC_TRY(grow(h, h->td_tau, (size_t)n * sizeof(double)));
  SC_TRY(grow(h, h->td_work, (size_t)(5 * (size_t)n + 2048) * sizeof(double)));
  if (is_lap) {
    launch_laplacian(s, M, scratch, n, ld, laplacian_type, ptr<double>(h->deg));
  } else {
    SC_HIP(h, hipMemcpyAsync(scratch, M, (size_t)n * ld * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  const double t_begin = sw::eig_trace() ? now_us() : 0.0;
  // a badly scaled matrix (max|a| beyond 2^+-400) is brought to [1, 2) by a power of two first:
  // the reflectors square their columns (hessenberg.hip); the eigenvalues get the factor back
  double eig_scale = 1.0;
  {
    SC_TRY(grow(h, h->fscal, 4 * sizeof(double)));
    SC_HIP(h, hipMemsetAsync(h->fscal.p, 0, sizeof(double), s));
    launch_free_absmax(s, scratch, n, ld, ptr<double>(h->fscal));
    double amax = 0.0;
    SC_HIP(h, hipMemcpyAsync(&amax, h->fscal.p, sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipStreamSynchronize(s));
    int e = 0;
    if (amax > 0.0 && std::isfinite(amax)) {
      std::frexp(amax, &e);  // amax = f * 2^e, f in [0.5, 1)
      if (e > 400 || e < -400) {
        eig_scale = std::ldexp(1.0, e - 1);  // amax / eig_scale in [1, 2)
        launch_scale_matrix(s, scratch, ld, n, std::ldexp(1.0, 1 - e));
      }
    }
  }
  SC_HIP(h, hipMemsetAsync(h->td_tau.p, 0, (size_t)n * sizeof(double), s));
  launch_hessenberg(s, scratch, ld, n, ptr<double>(h->td_tau), ptr<double>(h->td_work));
  SC_TRY(check_last(h, "Hessenberg reduction launch"));
  std::vector<double> packed((size_t)n * ld), tau(n);
  SC_HIP(h, hipMemcpyAsync(packed.data(), scratch, packed.size() * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(tau.data(), h->td_tau.p, (size_t)n * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(h->h_flags + 12, ptr<int>(h->flags) + 12, sizeof(int),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
  const double t_reduced = sw::eig_trace() ? now_us() : 0.0;
  HostHessenberg hw;
  if (!host_hessenberg_unpack(packed.data(), (size_t)ld, n, tau.data(), &hw))
    return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
  packed.clear();
  packed.shrink_to_fit();
  std::vector<double> wr(n), wi(n);
  if (!host_hessenberg_eigenvalues(hw, wr.data(), wi.data()))
    return fail(h, SC_ERR_NOT_CONVERGED, "QR iteration on the Hessenberg form failed");
  const double t_values = sw::eig_trace() ? now_us() : 0.0;
  // (hw, wr, wi stay in the scaled units for the inverse iteration below; theta -- what the
  //  eigengap reads and the caller gets -- is in the matrix's own)
  // np.linalg.eig + .real + argsort (utils.py:59-67): by real part, descending for the
  // affinity itself, ascending for a Laplacian (= descending in -L, the convention of `theta`)
  const double sign = is_lap ? -1.0 : 1.0;
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(),
                   [&](int a, int b) { return sign * wr[a] > sign * wr[b]; });
  std::vector<double> theta(n), zeros(n, 0.0);
  for (int i = 0; i < n; ++i) theta[i] = sign * wr[order[i]] * eig_scale;
  EigDecision dc = analyze(rq, theta.data(), zeros.data(), n, n, true, false);
  if (!dc.enough) return fail(h, SC_ERR_UNSUPPORTED, "eigen request cannot be satisfied");
  const int cols = std::max(1, std::min(n, dc.kvec));
  SC_TRY(ensure_vectors(h, n, cols));
  const int ldv = round_up(n, 16);
  SC_TRY(grow(h, h->Vre, (size_t)ldv * std::max(cols, kGenMax) * sizeof(double)));
  SC_TRY(grow(h, h->Vim, (size_t)ldv * std::max(cols, kGenMax) * sizeof(double)));
  std::vector<double> pr(cols), pi(cols), vre((size_t)ldv * cols, 0.0), vim((size_t)ldv * cols, 0.0);
  for (int q = 0; q < cols; ++q) {
    pr[q] = wr[order[q]];
    pi[q] = wi[order[q]];
  }
  double max_resid = 0.0;
  if (!host_hessenberg_vectors(hw, pr.data(), pi.data(), cols, vre.data(), vim.data(), (size_t)ldv,
                               &max_resid))
    return fail(h, SC_ERR_NOT_CONVERGED, "inverse iteration on the Hessenberg form failed");
  SC_HIP(h, hipMemcpyAsync(h->Vre.p, vre.data(), vre.size() * sizeof(double), hipMemcpyHostToDevice, s));
  SC_HIP(h, hipMemcpyAsync(h->Vim.p, vim.data(), vim.size() * sizeof(double), hipMemcpyHostToDevice, s));
  launch_gen_phase(s, ptr<double>(h->Vre), ptr<double>(h->Vim), ldv, n, cols, ptr<double>(h->E), ldv);
  SC_TRY(check_last(h, "eigenvector normalisation launch"));
  SC_HIP(h, hipStreamSynchronize(s));  // vre / vim are locals
  if (sw::eig_trace())
    fprintf(stderr, "[sc] dense general route n=%d: Hessenberg reduction (device, incl. the copy "
            "back) %.1f ms, %d eigenvalues by QR (host) %.1f ms, %d eigenvectors by inverse "
            "iteration + back-transform (host) and phase (device) %.1f ms\n", n,
            (t_reduced - t_begin) * 1e-3, n, (t_values - t_reduced) * 1e-3, cols,
            (now_us() - t_values) * 1e-3);
  h->n_vec = cols;
  dc.kw = n;
  dc.converged = true;
  dc.max_resid = max_resid * std::max(hw.norm, 1e-300) * eig_scale;
  if (out_w) {  // the whole spectrum, in the reference's order
    out_w->resize(n);
    for (int i = 0; i < n; ++i) (*out_w)[i] = rq.descend ? theta[i] : -theta[i];
  }
  if (diag) {
    diag->eig_path = SC_EIG_PATH_DENSE_HESSENBERG;
    diag->eig_matvec_passes = 0;
    diag->eig_block = kEigBlock;
    diag->eig_basis = n;
    diag->eig_cycles = 0;
    diag->eig_max_residual = dc.max_resid;
    diag->eig_fallback = reason;
  }
  *out_dc = dc;
  return SC_OK;
}

int gen_topk(sc_handle h, const double* M, int ld, int n, int laplacian_type,
                    const EigRequest& rq_in, sc_diag* diag, EigDecision* out_dc,
                    std::vector<double>* out_w, double* scratch) {
  EigRequest rq = rq_in;
  hipStream_t s = h->stream;
  SC_TRY(ensure_eig(h, n));
  SC_TRY(ensure_gen(h, n));
  double* theta_d = ptr<double>(h->theta);
  double* thetai_d = ptr<double>(h->thetai);
  double* resid_d = ptr<double>(h->resid);
  double* Yre = ptr<double>(h->Y);
  double* Yim = ptr<double>(h->Yt);
  double* Vre = ptr<double>(h->Vre);
  double* Vim = ptr<double>(h->Vim);
  int* info_d = ptr<int>(h->flags) + 8;
  const int ldv = round_up(n, 16);
  double* th = h->h_theta;             // [0, kLdq): Re theta, [kLdq, 2 kLdq): resid
  double* thi = h->h_theta + 2 * kLdq;  // Im theta
  EigDecision dc;
  int m = 0, passes = 0, cycles = 0;
  const bool is_lap = laplacian_type >= SC_LAPLACIAN_UNNORMALIZED;
  bool far_end = !rq.descend && rq.eigengap_type == SC_EIGENGAP_NORMALIZED_DIFF &&
                 rq.fixed_count == 0;
  int far_passes = 0, far_cycles = 0;

  auto fetch_ritz = [&](int count) -> int {
    SC_HIP(h, hipMemcpyAsync(th, theta_d, count * sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipMemcpyAsync(thi, thetai_d, count * sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipMemcpyAsync(h->h_flags + 8, info_d, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipMemcpyAsync(h->h_flags + 12, ptr<int>(h->flags) + 12, sizeof(int),
                             hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipStreamSynchronize(s));
    if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
    if (h->h_flags[8] != 0)
      return fail(h, SC_ERR_NOT_CONVERGED, "QR iteration of the projected eigenproblem failed");
    return SC_OK;
  };

  if (n <= kGenMax) {
    // ---- dense: eigen-decomposition of the reference's own matrix
    const double* src = M;
    if (is_lap) {
      launch_laplacian(s, M, ptr<double>(h->genL), n, ld, laplacian_type, ptr<double>(h->deg));
      src = ptr<double>(h->genL);
    }
    launch_gen_eig(s, src, ld, n, is_lap ? -1.0 : 1.0, n, theta_d, thetai_d, Yre, Yim, kLdq,
                   info_d);
    SC_TRY(check_last(h, "dense general eigensolver launch"));
    SC_TRY(fetch_ritz(n));
    for (int i = 0; i < n; ++i) th[kLdq + i] = 0.0;
    dc = analyze(rq, th, th + kLdq, n, n, true, false);
    if (!dc.enough) return fail(h, SC_ERR_UNSUPPORTED, "eigen request cannot be satisfied");
    launch_gen_ritz(s, nullptr, 0, n, n, Yre, Yim, kLdq, n, Vre, Vim, ldv);
    launch_gen_phase(s, Vre, Vim, ldv, n, n, ptr<double>(h->E), ldv);
    SC_TRY(check_last(h, "eigenvector normalisation launch"));
    h->n_vec = n;
    m = n;
    dc.kw = n;
    if (diag) diag->eig_path = SC_EIG_PATH_DENSE_GENERAL;
  } else {
    // test switch (tests/test_gpu_alternate_paths.py): straight to the landing pad
    if (sw::eig_force_dense())
      return gen_dense_large(h, M, ld, n, laplacian_type, rq, diag, out_dc, out_w, scratch, 4);
    // a small problem: the dense route costs 20-60 ms up to n = 512 and returns np.linalg.eig's
    // whole spectrum to rounding level -- block Arnoldi is faster there, but its decision-aware
    // stop (below) leaves the consumed values that cannot move the decision at 1e-3.  Parity
    // first where it is this cheap; fixed-count stage requests keep the Krylov solver.
    if (rq.fixed_count == 0 && n <= sw::gen_dense_max_n())
      return gen_dense_large(h, M, ld, n, laplacian_type, rq, diag, out_dc, out_w, scratch, 8);
    // max_clusters=None with a Laplacian reads every eigenvalue (utils.py:100-115): dense route
    if (rq.fixed_count == 0 && rq.max_clusters == 0 && !rq.descend)
      return gen_dense_large(h, M, ld, n, laplacian_type, rq, diag, out_dc, out_w, scratch, 6);
    // The ascending NormalizedDiff gap divides by np.max(eigenvalues) (utils.py:110,123): the FAR
    // end of the Laplacian's spectrum, the edge of a dense bulk where a Krylov space converges
    // like 1 / degree^2 -- 1e-4 .. 4e-3 on the far-end Ritz value of a 64-vector basis
    // (profiles/r18_general_strict_probe.txt), and max_delta inherits that error.  It is a
    // consumed eigenvalue like the others.
    // So it gets a solve of its own first: ONE eigenvalue, the one of largest real part of the
    // operator with its sign turned (+L), to value_tol by its residual; the main solve below then
    // divides by that value instead of its own far-end Ritz value.  Tens of block passes where the
    // dense route is seconds at n = 2000.  Only a far-end solve that spends its restart budget
    // sends the request to the dense route (eig_fallback 9).
    if (far_end && !sw::gen_loose_bulk()) {
      EigRequest fr = rq;
      fr.fixed_count = 1;
      fr.descend = 1;
      fr.negate = !rq.negate;
      fr.no_dense = 1;
      fr.decision_aware = 0;
      // (a residual bounds the error of an eigenvalue of a non-normal matrix only up to the
      //  eigenvalue's condition number: 7.8e-6 was seen at a residual of 1e-6,
      //  profiles/r25_general_far_end_probe.txt -- one decade of margin costs ~1/6 more passes)
      fr.value_tol = 0.1 * rq.value_tol;
      fr.vector_tol = fr.value_tol;  // (the vector is not used)
      fr.max_cycles = std::max(rq.max_cycles, 60);
      EigDecision fdc;
      std::vector<double> fw;
      sc_diag fdiag;
      memset(&fdiag, 0, sizeof(fdiag));
      const int rc_far = gen_topk(h, M, ld, n, laplacian_type, fr, &fdiag, &fdc, &fw, scratch);
      if (rc_far == SC_OK && !fw.empty()) {
        rq.have_far = 1;
        rq.far_value = fw[0];
        far_end = false;
        far_passes = fdiag.eig_matvec_passes;
        far_cycles = fdiag.eig_cycles;
        if (sw::eig_trace())
          fprintf(stderr, "[sc] general eigen path: far end %.12g in %d passes, %d cycles\n",
                  fw[0], far_passes, far_cycles);
      } else if (rc_far == SC_ERR_NOT_CONVERGED && n <= kGenDenseLimit) {
        return gen_dense_large(h, M, ld, n, laplacian_type, rq, diag, out_dc, out_w, scratch, 9);
      } else if (rc_far != SC_OK && rc_far != SC_ERR_NOT_CONVERGED) {
        return rc_far;
      }
      // (beyond the dense route's size limit: the far end as the main solve's basis has it)
    }
    // Narrow form: basis <= 64, projected problems solved by the one-wavefront device kernel,
    // up to 32 Ritz pairs.  WIDE form (a request for more -- max_clusters up to 63,
    // min_clusters up to 64 -- or a descending request whose stop_eigenvalue turns out to lie
    // deeper): basis <= 128, the projected problems (order <= 128) solved on the host
    // (host_general_eig), up to 64 pairs.
    const int asked = rq.fixed_count > 0
                          ? rq.fixed_count
                          : std::max(rq.max_clusters > 0 ? rq.max_clusters + 1 : 0, rq.min_clusters);
    // (the wide form for every request was measured in round 6: 2.5 ms per pass against 0.85 --
    //  a basis of 128 and host solves of order 128 cost more than they save in passes)
    bool wide = asked > 32;
    if (asked > 64)  // more pairs than the wide Arnoldi basis yields: dense route
      return gen_dense_large(h, M, ld, n, laplacian_type, rq, diag, out_dc, out_w, scratch, 5);
  general_restart:
    if (wide) {  // Ritz vectors: 104 kept + 1 far end + 8 residual columns
      SC_TRY(grow(h, h->Vre, (size_t)ldv * 2 * kGenMax * sizeof(double)));
      SC_TRY(grow(h, h->Vim, (size_t)ldv * 2 * kGenMax * sizeof(double)));
      Vre = ptr<double>(h->Vre);
      Vim = ptr<double>(h->Vim);
    }
    m = 0;
    cycles = 0;
    const double* cl = ptr<double>(h->cvec);
    const double* cr = ptr<double>(h->crvec);
    const double* pv = ptr<double>(h->pvec);
    if (rq.negate) {  // -Op x = (-p) .* x + (-cl) .* (M (cr .* x))
      const size_t stride = round_up(n, 16);
      SC_TRY(grow(h, h->gneg, 2 * stride * sizeof(double)));
      launch_negate2(s, cl, pv, n, ptr<double>(h->gneg), stride);
      cl = ptr<double>(h->gneg);
      pv = ptr<double>(h->gneg) + stride;
    }
    double* Q = ptr<double>(h->Q);
    double* OpQ = ptr<double>(h->Q2);
    double* W = ptr<double>(h->W);
    h->vs_scale = cr;
    struct Restore {
      sc_handle h;
      ~Restore() { h->vs_scale = nullptr; }
    } restore{h};
    const int cap = std::min(wide ? kEigBasisCap : kGenMax, ((n - kEigBlock) / kEigBlock) * kEigBlock);
    const int first_check = std::min(3 * kEigBlock, cap);
    uint64_t seed = 0x9e3779b97f4a7c15ull;
    std::vector<std::vector<int>> start_blocks;  // restart: codes 2*col+part, -1 = noise
    size_t next_start = 0;
    const int kMaxCheck = wide ? 64 : 32;  // Ritz pairs whose residual is evaluated
    std::vector<double> hy;  // wide: host copies of the Ritz coefficient vectors (re | im)
    while (true) {
      // ---- next block into W
      if (next_start < start_blocks.size()) {
        SC_HIP(h, hipMemcpyAsync(h->gsrc.p, start_blocks[next_start].data(),
                                 kEigBlock * sizeof(int), hipMemcpyHostToDevice, s));
        launch_gen_gather(s, Vre, Vim, ldv, n, ptr<int>(h->gsrc), ++seed, W);
        SC_HIP(h, hipStreamSynchronize(s));  // the code vector is host memory
        ++next_start;
      } else if (m == 0) {
        launch_random_block(s, W, n, seed);
      } else {
        launch_copy_block(s, OpQ + (m - kEigBlock), kLdq, W, kEigBlock, n, kEigBlock);
      }
      SC_TRY(orthonormalize(h, n, m, false, 0, m, false));
      SC_TRY(finish_block(h, n, m, m, &seed));
      launch_block_matvec(s, M, ld, n, cl, pv, Q + m, kLdq, ptr<double>(h->Vs), W);
      launch_copy_block(s, W, kEigBlock, OpQ + m, kLdq, n, kEigBlock);
      ++passes;
      m += kEigBlock;
      // Rayleigh-Ritz (a serial ~m^3 solve in one wavefront) is the expensive step: every
      // block early in the first cycle, where convergence is expected, then every other
      // block, and only with a full basis once restarts have begun
      const bool check = next_start >= start_blocks.size() && m >= first_check &&
                         (cycles == 0 ? (m <= 4 * kEigBlock || m % (2 * kEigBlock) == 0 ||
                                         m + kEigBlock > cap)
                                      : (m + kEigBlock > cap));
      if (check) {
        // H = Q^T (Op Q), one 8-column block at a time
        for (int jb = 0; jb < m; jb += kEigBlock) {
          launch_copy_block(s, OpQ + jb, kLdq, W, kEigBlock, n, kEigBlock);
          launch_proj_partial(s, Q, kLdq, m, W, n, ptr<double>(h->partial));
          launch_reduce_H(s, ptr<double>(h->partial), proj_blocks(n), m, ptr<double>(h->Hbuf),
                          nullptr, 0, 0, 0, ptr<double>(h->hsq));
          launch_copy_block(s, ptr<double>(h->Hbuf), kEigBlock, ptr<double>(h->T) + jb, kLdq,
                            m, kEigBlock);
        }
        if (!wide && sw::gen_device_rr()) {
          launch_gen_eig(s, ptr<double>(h->T), kLdq, m, 1.0, m, theta_d, thetai_d, Yre, Yim, kLdq,
                         info_d);
        } else {
          // The projected problem (order <= 64 narrow, <= 128 wide) is solved on the HOST: T comes
          // over, its eigenpairs go back to where k_gen_eig leaves them.  Round 6: in the narrow
          // form too -- the one-wavefront device kernel takes 4.0 ms at m = 64 and was 77 % of the
          // GPU time of a general-path call (profiles/r35_gen_kernel_stats.txt); real double-shift
          // QR + inverse iteration for the vectors a restart keeps take 0.3-0.9 ms
          // (host_general_eig_fast; the complex Schur form is its fallback).  SC_GEN_DEVICE_RR=1:
          // the device kernel (narrow form).
          SC_HIP(h, hipMemcpy2DAsync(h->h_rr, (size_t)m * sizeof(double), h->T.p,
                                     (size_t)kLdq * sizeof(double), (size_t)m * sizeof(double), m,
                                     hipMemcpyDeviceToHost, s));
          SC_HIP(h, hipStreamSynchronize(s));
          hy.assign(2 * (size_t)m * m, 0.0);
          for (size_t e = 0; e < (size_t)m * m; ++e)
            if (!std::isfinite(h->h_rr[e])) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
          // (vectors: the pairs whose residual is evaluated and a restart materialises; all of
          //  them while the far end -- Ritz pair m - 1 -- is tracked)
          const int nvec = far_end ? m : std::min(m, wide ? 104 : 40);
          if (!host_general_eig_fast(h->h_rr, m, m, nvec, th, thi, hy.data(),
                                     hy.data() + (size_t)m * m, m) &&
              !host_general_eig(h->h_rr, m, m, m, th, thi, hy.data(), hy.data() + (size_t)m * m, m))
            return fail(h, SC_ERR_NOT_CONVERGED, "QR iteration of the projected eigenproblem failed");
          SC_HIP(h, hipMemcpyAsync(theta_d, th, m * sizeof(double), hipMemcpyHostToDevice, s));
          SC_HIP(h, hipMemcpyAsync(thetai_d, thi, m * sizeof(double), hipMemcpyHostToDevice, s));
          SC_HIP(h, hipMemcpy2DAsync(Yre, (size_t)kLdq * sizeof(double), hy.data(),
                                     (size_t)m * sizeof(double), (size_t)m * sizeof(double), m,
                                     hipMemcpyHostToDevice, s));
          SC_HIP(h, hipMemcpy2DAsync(Yim, (size_t)kLdq * sizeof(double), hy.data() + (size_t)m * m,
                                     (size_t)m * sizeof(double), (size_t)m * sizeof(double), m,
                                     hipMemcpyHostToDevice, s));
          SC_HIP(h, hipMemsetAsync(info_d, 0, 2 * sizeof(int), s));
          SC_HIP(h, hipStreamSynchronize(s));  // (th / thi / hy are reused below)
        }
        const int c1 = std::min(m, kMaxCheck);
        for (int c0 = 0; c0 < c1; c0 += 32)  // (the residual kernel takes 32 pairs per launch)
          launch_gen_residual(s, Q, OpQ, kLdq, m, n, Yre + c0, Yim + c0, kLdq, theta_d + c0,
                              thetai_d + c0, std::min(32, c1 - c0), ptr<double>(h->gpart),
                              resid_d + c0);
        if (far_end && m - 1 >= c1)
          launch_gen_residual(s, Q, OpQ, kLdq, m, n, Yre + (m - 1), Yim + (m - 1), kLdq,
                              theta_d + (m - 1), thetai_d + (m - 1), 1, ptr<double>(h->gpart),
                              resid_d + (m - 1));
        SC_TRY(check_last(h, "Rayleigh-Ritz launch"));
        for (int i = 0; i < m; ++i) th[kLdq + i] = 1e300;  // not evaluated = not converged
        SC_HIP(h, hipMemcpyAsync(th + kLdq, resid_d, c1 * sizeof(double), hipMemcpyDeviceToHost,
                                 s));
        if (far_end && m - 1 >= c1)
          SC_HIP(h, hipMemcpyAsync(th + kLdq + m - 1, resid_d + (m - 1), sizeof(double),
                                   hipMemcpyDeviceToHost, s));
        SC_TRY(fetch_ritz(m));
        dc = analyze(rq, th, th + kLdq, m, n, false, false);
        if (sw::eig_trace())
          fprintf(stderr, "[sc] arnoldi pass %d m=%d cycle %d sweeps %d: enough=%d conv=%d kw=%d "
                  "kvec=%d fail kind %d at %d (resid %.2e)\n", passes, m, cycles, h->h_flags[9],
                  dc.enough, dc.converged, dc.kw, dc.kvec, dc.fail_kind, dc.fail_index,
                  dc.fail_index >= 0 ? th[kLdq + dc.fail_index] : 0.0);
        if (sw::eig_trace() > 1) {
          for (int i = 0; i < std::min(m, 12); ++i)
            fprintf(stderr, "[sc]    ritz %2d  re %.12g  im %.3e  resid %.3e\n", i, th[i], thi[i],
                    th[kLdq + i]);
        }
        if (!wide && !dc.unsupported && (!dc.enough ? m >= kMaxCheck && rq.max_clusters == 0 && rq.fixed_count == 0
                                                    : std::max(dc.kw, dc.kvec) > kMaxCheck)) {
          // a descending request that reads further than 32 values: once more, wide
          if (sw::eig_trace()) fprintf(stderr, "[sc] arnoldi: more than 32 pairs wanted, wide form\n");
          wide = true;
          passes = 0;
          goto general_restart;
        }
        // (a descending request whose stop_eigenvalue lies deeper than 64 values: dense route)
        if (dc.unsupported || (dc.enough && std::max(dc.kw, dc.kvec) > kMaxCheck))
          return gen_dense_large(h, M, ld, n, laplacian_type, rq, diag, out_dc, out_w, scratch, 5);
        if (dc.enough && dc.converged) break;
      }
      if (m + kEigBlock > cap) {
        // ---- explicit restart from the wanted Ritz vectors
        if (++cycles > rq.max_cycles) {  // restart budget spent: the landing pad
          if (rq.no_dense)
            return fail(h, SC_ERR_NOT_CONVERGED, "far-end eigenvalue: restart budget spent");
          return gen_dense_large(h, M, ld, n, laplacian_type, rq, diag, out_dc, out_w, scratch, 1);
        }
        // (thick restart: the new basis is [wanted Ritz vectors | residual block], after
        // which the Arnoldi recurrence continues from the residual block)
        const int kStash = wide ? 112 : 48;  // Vre columns [kStash, kStash + 8): the residual block
        launch_copy_block(s, OpQ + (m - kEigBlock), kLdq, W, kEigBlock, n, kEigBlock);
        SC_TRY(orthonormalize(h, n, m, false, 0, -1, false));
        SC_TRY(finish_block(h, n, m, -1, &seed));
        launch_rowmajor_to_colmajor(s, W, kEigBlock, n, kEigBlock, Vre + (size_t)kStash * ldv,
                                    ldv);
        const int want = dc.enough ? std::max(dc.kw, dc.kvec) : cap / 4;
        const int avail = std::min(m, wide ? 104 : 40);  // Ritz vectors materialised: columns [0, avail)
        launch_gen_ritz(s, Q, kLdq, m, n, Yre, Yim, kLdq, avail, Vre, Vim, ldv);
        int vcols = avail;
        const bool far_kept = far_end && m - 1 >= avail;
        if (far_kept) {  // Ritz vector m-1 -> column `avail`
          launch_gen_ritz(s, Q, kLdq, m, n, Yre + (m - 1), Yim + (m - 1), kLdq, 1,
                          Vre + (size_t)avail * ldv, Vim + (size_t)avail * ldv, ldv);
          vcols = avail + 1;
        }
        launch_gen_phase(s, Vre, Vim, ldv, n, vcols, nullptr, 0);
        SC_TRY(check_last(h, "restart launch"));
        const double scale = std::max(std::fabs(th[0]), std::fabs(th[m - 1]));
        auto is_complex = [&](int i) { return std::fabs(thi[i]) > 1e-12 * std::max(scale, 1e-300); };
        std::vector<int> codes;
        if (far_kept) {
          codes.push_back(2 * avail);
          if (is_complex(m - 1)) codes.push_back(2 * avail + 1);
        }
        // Kept vectors must fill whole blocks: a random pad column r would break the
        // relation Op [kept] in span(kept, residual block) -- (I - Q Q^T) Op r is not in the
        // basis -- and the Ritz pairs then stall at the size of their component along r.
        // So the kept set is extended, never padded.
        const int max_cols = (std::max(1, cap / kEigBlock - 3)) * kEigBlock;
        const int target = std::min(round_up(want + kEigBlock / 2, kEigBlock), max_cols);
        for (int i = 0; i < avail; ++i) {
          if ((int)codes.size() >= target && codes.size() % kEigBlock == 0) break;
          if (!is_complex(i)) {
            codes.push_back(2 * i);
            continue;
          }
          bool partner_kept = false;  // its conjugate, earlier in the list
          for (int j = 0; j < i; ++j)
            if (is_complex(j) && std::fabs(th[j] - th[i]) <= 1e-9 * scale &&
                std::fabs(thi[j] + thi[i]) <= 1e-9 * scale)
              partner_kept = true;
          if (partner_kept) continue;
          codes.push_back(2 * i);
          codes.push_back(2 * i + 1);
        }
        if ((int)codes.size() > max_cols) codes.resize(max_cols);
        while (codes.size() % kEigBlock) codes.push_back(-1);  // last resort (tiny bases)
        for (int j = 0; j < kEigBlock; ++j) codes.push_back(2 * (kStash + j));
        start_blocks.clear();
        for (size_t b = 0; b * kEigBlock < codes.size(); ++b)
          start_blocks.emplace_back(codes.begin() + b * kEigBlock,
                                    codes.begin() + (b + 1) * kEigBlock);
        next_start = 0;
        m = 0;
      }
    }
    const int cols = std::min(std::max(dc.kw, dc.kvec), kMaxCheck);
    launch_gen_ritz(s, Q, kLdq, m, n, Yre, Yim, kLdq, cols, Vre, Vim, ldv);
    launch_gen_phase(s, Vre, Vim, ldv, n, cols, ptr<double>(h->E), ldv);
    SC_TRY(check_last(h, "ritz vector launch"));
    h->n_vec = cols;
    if (diag) diag->eig_path = SC_EIG_PATH_BLOCK_ARNOLDI;
  }
  if (out_w) {
    out_w->resize(dc.kw);
    for (int i = 0; i < dc.kw; ++i) (*out_w)[i] = rq.descend ? th[i] : -th[i];
  }
  if (diag) {
    diag->eig_matvec_passes = passes + far_passes;  // (the far-end solve's count too)
    diag->eig_block = kEigBlock;
    diag->eig_basis = m;
    diag->eig_cycles = cycles + far_cycles;
    diag->eig_max_residual = dc.max_resid;
  }
  *out_dc = dc;
  return SC_OK;
}

