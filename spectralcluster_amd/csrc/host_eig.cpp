// Host-side dense kernels of the eigen drivers: see host_eig.h.  Textbook algorithms
// (EISPACK tred2 / tql2 / tql1, LAPACK dsytd2 / dstein), none of them from the reference,
// which calls numpy.linalg.eig (utils.py:59).
#include "host_eig.h"

#include <algorithm>
#include <thread>
#include <mutex>
#include <atomic>
#include <cmath>
#include <cstdint>

#include "../../include/spectralcluster_amd.h"

// ------------------------------------------------------------------------------
// Rayleigh-Ritz on the host: full solve of small projected problems (m <= kHostRR = 64)
// ------------------------------------------------------------------------------
// The projected matrix T = Q^T Op Q of the first checks is 24 x 24 .. 48 x 48: 4.6-18 KB that
// come back with the flags the host reads anyway.  A one-workgroup Jacobi takes 170-330 us
// for it on the device (a chain of ~160 barrier-separated rounds); Householder
// tridiagonalisation + implicit QL (the textbook tred2 / tql2 recurrences) on one host core
// takes ~20-60 us.  Larger bases (clustered spectra, restarts) take the leading-vector
// solve further down (host_partial_*); the device Jacobi remains behind SC_EIG_DEVICE_RR.
//
// a: m x m symmetric (row-major, lda), overwritten with the eigenvectors (columns);
// d: eigenvalues ascending.  Returns false if QL did not converge (30 iterations).
bool host_symmetric_eig(double* a, int lda, int m, double* d, double* e) {
  auto A = [&](int i, int j) -> double& { return a[(size_t)i * lda + j]; };
  // ---- tred2: Householder reduction to tridiagonal form, accumulating the transformation
  for (int i = m - 1; i >= 1; --i) {
    const int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale += std::fabs(A(i, k));
      if (scale == 0.0) {
        e[i] = A(i, l);
      } else {
        for (int k = 0; k <= l; ++k) {
          A(i, k) /= scale;
          h += A(i, k) * A(i, k);
        }
        double f = A(i, l);
        double g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
        e[i] = scale * g;
        h -= f * g;
        A(i, l) = f - g;
        f = 0.0;
        for (int j = 0; j <= l; ++j) {
          A(j, i) = A(i, j) / h;
          g = 0.0;
          for (int k = 0; k <= j; ++k) g += A(j, k) * A(i, k);
          for (int k = j + 1; k <= l; ++k) g += A(k, j) * A(i, k);
          e[j] = g / h;
          f += e[j] * A(i, j);
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) {
          f = A(i, j);
          e[j] = g = e[j] - hh * f;
          for (int k = 0; k <= j; ++k) A(j, k) -= (f * e[k] + g * A(i, k));
        }
      }
    } else {
      e[i] = A(i, l);
    }
    d[i] = h;
  }
  d[0] = 0.0;
  e[0] = 0.0;
  for (int i = 0; i < m; ++i) {
    const int l = i - 1;
    if (d[i] != 0.0) {
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
        for (int k = 0; k <= l; ++k) g += A(i, k) * A(k, j);
        for (int k = 0; k <= l; ++k) A(k, j) -= g * A(k, i);
      }
    }
    d[i] = A(i, i);
    A(i, i) = 1.0;
    for (int j = 0; j <= l; ++j) A(j, i) = A(i, j) = 0.0;
  }
  // ---- tql2: implicit QL with eigenvector accumulation
  for (int i = 1; i < m; ++i) e[i - 1] = e[i];
  e[m - 1] = 0.0;
  for (int l = 0; l < m; ++l) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < m - 1; ++mm) {
        const double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
        if (std::fabs(e[mm]) <= 2.220446049250313e-16 * dd) break;
      }
      if (mm != l) {
        if (iter++ == 60) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[mm] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          e[i + 1] = r = std::hypot(f, g);
          if (r == 0.0) {
            d[i + 1] -= p;
            e[mm] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          for (int k = 0; k < m; ++k) {
            f = A(k, i + 1);
            A(k, i + 1) = s * A(k, i) + c * f;
            A(k, i) = c * A(k, i) - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[mm] = 0.0;
      }
    } while (mm != l);
  }
  return true;
}

// host-only export: lets the CPU tests pin the routine without a GPU
extern "C" int sc_host_symmetric_eig(const double* a, int m, double* values, double* vectors) {
  if (!a || !values || !vectors || m < 1) return SC_ERR_INVALID;
  std::vector<double> e(m);
  for (size_t i = 0; i < (size_t)m * m; ++i) vectors[i] = a[i];
  return host_symmetric_eig(vectors, m, m, values, e.data()) ? SC_OK : SC_ERR_NOT_CONVERGED;
}

// ------------------------------------------------------------------------------
// eigenvectors of a symmetric tridiagonal matrix by inverse iteration (host)
// ------------------------------------------------------------------------------
// LAPACK dstein's method for a handful of eigenvalues `lam` (any order; neighbours in the
// list closer than 1e-3 ||T|| are treated as a cluster and kept orthogonal by modified
// Gram-Schmidt, exact duplicates are separated by 10 ulp like dstein's `pertol`): LU of
// T - lam I with partial pivoting (dlagtf's elimination), a random start, solves until
// the iterate has grown past dstein's threshold plus two more.  O(n) per vector and
// iteration -- a serial recurrence, which is why it runs here and not on the device; the
// O(n^2) back-transform is k_td_backtransform.  Z: column-major, column q at Z + q * ldz,
// unit 2-norm, largest component positive.  Returns false if a vector failed to grow.
bool host_tridiag_eigvectors(const double* d, const double* e, int n, const double* lam,
                                    int k, double* Z, size_t ldz) {
  if (n == 1) {
    for (int q = 0; q < k; ++q) Z[q * ldz] = 1.0;
    return true;
  }
  const double eps = 2.220446049250313e-16;
  double onenrm = std::fabs(d[0]) + std::fabs(e[0]);
  onenrm = std::max(onenrm, std::fabs(d[n - 1]) + std::fabs(e[n - 2]));
  for (int i = 1; i < n - 1; ++i)
    onenrm = std::max(onenrm, std::fabs(d[i]) + std::fabs(e[i - 1]) + std::fabs(e[i]));
  if (!(onenrm > 0.0)) onenrm = 1.0;
  const double ortol = 1e-3 * onenrm;
  const double pivtol = eps * onenrm;
  const double grow = std::sqrt(0.1 / n);  // dstein's dtpcrt
  std::vector<double> u0(n), u1(n), u2(n), l(n), x(n);
  std::vector<char> piv(n);
  std::vector<double> used(k);
  uint64_t rng = 0x9e3779b97f4a7c15ull;
  bool all_ok = true;
  int cluster_begin = 0;
  for (int q = 0; q < k; ++q) {
    double xj = lam[q];
    if (q > 0 && std::fabs(lam[q] - lam[q - 1]) >= ortol) cluster_begin = q;
    // separate (numerically) repeated shifts inside a cluster, keeping the list's direction
    for (int r = cluster_begin; r < q; ++r) {
      const double pert = 10.0 * eps * std::max(std::fabs(xj), onenrm * 1e-3);
      if (std::fabs(xj - used[r]) < pert) xj = used[r] + (lam[q] <= lam[cluster_begin] ? -pert : pert);
    }
    used[q] = xj;
    // ---- P L U = T - xj I  (row i of U: u0 diagonal, u1, u2 superdiagonals)
    double a = d[0] - xj;       // current diagonal entry of the row being eliminated with
    double b = n > 1 ? e[0] : 0.0;  // its first superdiagonal
    for (int i = 0; i < n - 1; ++i) {
      const double c = e[i];                      // subdiagonal entry (i + 1, i)
      const double an = d[i + 1] - xj;            // row i + 1: (c, an, bn)
      const double bn = i + 2 < n ? e[i + 1] : 0.0;
      if (std::fabs(a) >= std::fabs(c)) {         // no interchange
        double pv = a;
        if (std::fabs(pv) < pivtol) pv = pv < 0.0 ? -pivtol : pivtol;
        const double m = c / pv;
        piv[i] = 0; l[i] = m;
        u0[i] = pv; u1[i] = b; u2[i] = 0.0;
        a = an - m * b;
        b = bn;
      } else {                                    // rows i and i + 1 swapped
        const double m = a / c;
        piv[i] = 1; l[i] = m;
        u0[i] = c; u1[i] = an; u2[i] = bn;
        a = b - m * an;
        b = -m * bn;
      }
    }
    if (std::fabs(a) < pivtol) a = a < 0.0 ? -pivtol : pivtol;
    u0[n - 1] = a; u1[n - 1] = 0.0; u2[n - 1] = 0.0;
    // ---- inverse iteration
    for (int i = 0; i < n; ++i) {
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      x[i] = (double)(int64_t)(rng >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    }
    double* z = Z + (size_t)q * ldz;
    int extra = 0;
    bool ok = false;
    for (int it = 0; it < 12; ++it) {
      // scale the right-hand side to norm ~ n * onenrm * eps-ish (dstein) so that the solve
      // of a nearly singular system cannot overflow
      double amax = 0.0;
      for (int i = 0; i < n; ++i) amax = std::max(amax, std::fabs(x[i]));
      if (!(amax > 0.0)) { x[0] = 1.0; amax = 1.0; }
      const double scl = n * onenrm * std::max(eps, std::fabs(u0[n - 1])) / amax;
      for (int i = 0; i < n; ++i) x[i] *= scl;
      // L^-1 P
      for (int i = 0; i < n - 1; ++i) {
        if (piv[i]) {
          const double t = x[i];
          x[i] = x[i + 1];
          x[i + 1] = t - l[i] * x[i];
        } else {
          x[i + 1] -= l[i] * x[i];
        }
      }
      // U^-1
      x[n - 1] /= u0[n - 1];
      if (n > 1) x[n - 2] = (x[n - 2] - u1[n - 2] * x[n - 1]) / u0[n - 2];
      for (int i = n - 3; i >= 0; --i)
        x[i] = (x[i] - u1[i] * x[i + 1] - u2[i] * x[i + 2]) / u0[i];
      // keep the cluster orthogonal
      for (int r = cluster_begin; r < q; ++r) {
        const double* zr = Z + (size_t)r * ldz;
        double dot = 0.0;
        for (int i = 0; i < n; ++i) dot += x[i] * zr[i];
        for (int i = 0; i < n; ++i) x[i] -= dot * zr[i];
      }
      double nrm = 0.0;
      for (int i = 0; i < n; ++i) nrm = std::max(nrm, std::fabs(x[i]));
      if (!std::isfinite(nrm)) break;
      if (nrm >= grow) {
        if (++extra > 2) { ok = true; break; }
      }
    }
    double s2 = 0.0;
    int imax = 0;
    for (int i = 0; i < n; ++i) {
      s2 += x[i] * x[i];
      if (std::fabs(x[i]) > std::fabs(x[imax])) imax = i;
    }
    if (!(s2 > 0.0) || !std::isfinite(s2)) {
      ok = false;
      for (int i = 0; i < n; ++i) z[i] = 0.0;
    } else {
      const double inv = (x[imax] < 0.0 ? -1.0 : 1.0) / std::sqrt(s2);
      for (int i = 0; i < n; ++i) z[i] = x[i] * inv;
    }
    all_ok = all_ok && ok;
  }
  return all_ok;
}

// host-only export: lets the CPU tests pin the routine without a GPU.  vectors: (n, k)
// row-major (column q = eigenvector of lam[q]).
extern "C" int sc_host_tridiag_eigvectors(const double* d, const double* e, int n,
                                          const double* lam, int k, double* vectors) {
  if (!d || (!e && n > 1) || !lam || !vectors || n < 1 || k < 1) return SC_ERR_INVALID;
  std::vector<double> z((size_t)n * k);
  const bool ok = host_tridiag_eigvectors(d, e, n, lam, k, z.data(), (size_t)n);
  for (int q = 0; q < k; ++q)
    for (int i = 0; i < n; ++i) vectors[(size_t)i * k + q] = z[(size_t)q * n + i];
  return ok ? SC_OK : SC_ERR_NOT_CONVERGED;
}

// ------------------------------------------------------------------------------
// partial symmetric eigensolver on the host: ALL eigenvalues, the LEADING eigenvectors
// ------------------------------------------------------------------------------
// A Rayleigh-Ritz check of a basis of 72..128 vectors needs every Ritz value (the analysis
// reads the far end too) but only the leading max_clusters + 1 (+ one block, for a restart)
// Ritz VECTORS.  tred2 + tql2 above accumulate all m vectors: ~9 m^3 flops, 2 ms at m = 128.
// Here: Householder tridiagonalisation with the reflectors kept (LAPACK dsytd2, 4/3 m^3),
// eigenvalues by implicit QL without vectors (O(m^2)), the leading `need` vectors by
// inverse iteration on the tridiagonal form (host_tridiag_eigvectors) and Q z through the
// reflectors (O(m^2) each).

// a (m x m symmetric, FULL storage, row-major lda; destroyed) -> d, e (e[i] couples i, i+1),
// tau; reflector i: v(i, i+1) = 1, v(i, i+2 ..).
// Round 5: ONE pass over the trailing block per step.  The rank-2 update of step i
// (A22 -= v w^T + w v^T) and the matrix-vector product of step i + 1 (p' = tau' A22' v') touch
// the same rows: row c0 is updated first (it IS the next step's x, so v' and tau' follow), then
// every other row is updated and dotted with v' while it is in registers.  The dot products
// keep four independent partial sums (the compiler does not reassociate a floating-point
// reduction, so a single accumulator stays scalar).  (madd is a plain a * b + c, not an explicit
// fma: `target_clones` splits "avx2,fma" into an avx2 clone WITHOUT fma and an fma clone and the
// resolver prefers avx2 -- a __builtin_fma in it is a libm call, 6x slower than the code this
// replaced.)  Only the part of a row right of the next diagonal is kept up
// to date (what is left of it is never read again).
namespace {
inline double madd(double a, double b, double c) { return a * b + c; }
// LAPACK dlarfg on x[0 .. len): v[0] = 1, v[1 ..] = x[1 ..] / (alpha - beta); returns tau, *beta
__attribute__((target_clones("avx2,fma", "default")))
double make_reflector(const double* x, int len, double* v, double* beta_out) {
  const double alpha = x[0];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = 1;
  for (; k + 3 < len; k += 4) {
    s0 = madd(x[k], x[k], s0);
    s1 = madd(x[k + 1], x[k + 1], s1);
    s2 = madd(x[k + 2], x[k + 2], s2);
    s3 = madd(x[k + 3], x[k + 3], s3);
  }
  for (; k < len; ++k) s0 = madd(x[k], x[k], s0);
  const double xnorm2 = (s0 + s1) + (s2 + s3);
  if (!(xnorm2 > 0.0)) {
    for (int q = 0; q < len; ++q) v[q] = 0.0;
    *beta_out = alpha;
    return 0.0;
  }
  const double beta = -std::copysign(std::sqrt(alpha * alpha + xnorm2), alpha);
  const double scale = 1.0 / (alpha - beta);
  v[0] = 1.0;
  for (int q = 1; q < len; ++q) v[q] = x[q] * scale;
  *beta_out = beta;
  return (beta - alpha) / beta;
}
}  // namespace

__attribute__((target_clones("avx2,fma", "default")))
static void host_tridiagonalize(HostTridiag* w) {
  const int m = w->m, lda = w->lda;
  double* a = w->a.data();
  w->v.assign((size_t)m * lda, 0.0);
  double* vv = w->v.data();
  w->d.assign(m, 0.0);
  w->e.assign(std::max(m, 1), 0.0);
  w->tau.assign(std::max(m, 1), 0.0);
  if (m == 0) return;
  std::vector<double> pbuf(m), pnext(m), wbuf(m);
  double* p = pbuf.data();
  double* pn = pnext.data();
  double* ww = wbuf.data();
  // step 0: reflector of row 0 and p = tau A22 v by a plain product
  double tau = 0.0, beta = 0.0;
  if (m > 1) {
    const int len = m - 1;
    double* v = vv + 1;
    tau = make_reflector(a + 1, len, v, &beta);
    for (int r = 0; r < len; ++r) {
      const double* row = a + (size_t)(1 + r) * lda + 1;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int c = 0;
      for (; c + 3 < len; c += 4) {
        s0 = madd(row[c], v[c], s0);
        s1 = madd(row[c + 1], v[c + 1], s1);
        s2 = madd(row[c + 2], v[c + 2], s2);
        s3 = madd(row[c + 3], v[c + 3], s3);
      }
      for (; c < len; ++c) s0 = madd(row[c], v[c], s0);
      p[r] = tau * ((s0 + s1) + (s2 + s3));
    }
  }
  for (int i = 0; i + 1 < m; ++i) {
    const int c0 = i + 1, len = m - c0;
    const double* v = vv + (size_t)i * lda + c0;  // reflector i (len entries, v[0] = 1 or all 0)
    w->e[i] = beta;
    w->tau[i] = tau;
    w->d[i] = a[(size_t)i * lda + i];
    if (len == 1) {  // last step: a 1 x 1 trailing block, H = I - tau (scalar)
      if (tau != 0.0) {  // A22 <- (1 - tau)^2 A22: with the w form below
        const double dot = p[0] * v[0];
        const double w0 = p[0] - 0.5 * tau * dot * v[0];
        a[(size_t)c0 * lda + c0] -= 2.0 * v[0] * w0;
      }
      break;
    }
    // w = p - (tau / 2)(p^T v) v
    if (tau != 0.0) {
      double dot = 0.0;
      for (int r = 0; r < len; ++r) dot = madd(p[r], v[r], dot);
      const double half = 0.5 * tau * dot;
      for (int r = 0; r < len; ++r) ww[r] = madd(-half, v[r], p[r]);
    } else {
      for (int r = 0; r < len; ++r) ww[r] = 0.0;
    }
    // row c0 first: it carries the next step's x
    double* row0 = a + (size_t)c0 * lda + c0;
    if (tau != 0.0) {
      const double v0 = v[0], w0 = ww[0];
      for (int c = 0; c < len; ++c)
        row0[c] = madd(-v0, ww[c], madd(-w0, v[c], row0[c]));
    }
    double* vn = vv + (size_t)c0 * lda + (c0 + 1);  // reflector i + 1 (len - 1 entries)
    double beta_n = 0.0;
    const double tau_n = make_reflector(row0 + 1, len - 1, vn, &beta_n);
    // the other rows: update (columns >= 1 of the block) + product with the next reflector
    const int ln = len - 1;
    for (int r = 1; r < len; ++r) {
      double* row = a + (size_t)(c0 + r) * lda + (c0 + 1);
      const double vr = tau != 0.0 ? v[r] : 0.0, wr = ww[r];
      const double* vc = v + 1;
      const double* wc = ww + 1;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int c = 0;
      for (; c + 3 < ln; c += 4) {
        const double t0 = madd(-vr, wc[c], madd(-wr, vc[c], row[c]));
        const double t1 = madd(-vr, wc[c + 1], madd(-wr, vc[c + 1], row[c + 1]));
        const double t2 = madd(-vr, wc[c + 2], madd(-wr, vc[c + 2], row[c + 2]));
        const double t3 = madd(-vr, wc[c + 3], madd(-wr, vc[c + 3], row[c + 3]));
        row[c] = t0;
        row[c + 1] = t1;
        row[c + 2] = t2;
        row[c + 3] = t3;
        s0 = madd(t0, vn[c], s0);
        s1 = madd(t1, vn[c + 1], s1);
        s2 = madd(t2, vn[c + 2], s2);
        s3 = madd(t3, vn[c + 3], s3);
      }
      for (; c < ln; ++c) {
        const double t = madd(-vr, wc[c], madd(-wr, vc[c], row[c]));
        row[c] = t;
        s0 = madd(t, vn[c], s0);
      }
      pn[r - 1] = tau_n * ((s0 + s1) + (s2 + s3));
    }
    std::swap(p, pn);
    tau = tau_n;
    beta = beta_n;
  }
  w->d[m - 1] = a[(size_t)(m - 1) * lda + (m - 1)];
}

// eigenvalues of the tridiagonal (d, e) by implicit QL (tql1); ascending on return.
// (plain sqrt instead of hypot: the projected matrices are O(||Op||), nowhere near the
//  overflow range, and hypot was 3/4 of this routine's time)
static bool host_tridiag_values(std::vector<double> d, std::vector<double> e, int m,
                                std::vector<double>* out) {
  e.resize(m + 1, 0.0);
  e[m - 1] = 0.0;
  for (int l = 0; l < m; ++l) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < m - 1; ++mm) {
        const double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
        if (std::fabs(e[mm]) <= 2.220446049250313e-16 * dd) break;
      }
      if (mm != l) {
        if (iter++ == 60) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::sqrt(g * g + 1.0);
        g = d[mm] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          e[i + 1] = r = std::sqrt(f * f + g * g);
          if (r == 0.0) {
            d[i + 1] -= p;
            e[mm] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[mm] = 0.0;
      }
    } while (mm != l);
  }
  std::sort(d.begin(), d.begin() + m);
  out->assign(d.begin(), d.begin() + m);
  return true;
}

// Step 1: every eigenvalue of the m x m symmetric T (upper triangle given, row-major ld),
// DESCENDING into w->theta.
bool host_partial_values(const double* T, int ld, int m, HostTridiag* w) {
  w->m = m;
  w->lda = m | 1;
  w->a.assign((size_t)m * w->lda, 0.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j)  // the mirrored upper triangle is what the chain wrote
      w->a[(size_t)i * w->lda + j] = i <= j ? T[(size_t)i * ld + j] : T[(size_t)j * ld + i];
  host_tridiagonalize(w);
  std::vector<double> asc;
  if (!host_tridiag_values(w->d, w->e, m, &asc)) return false;
  w->theta.assign(asc.rbegin(), asc.rend());
  return true;
}

// Step 2: eigenvectors of the leading `need` eigenvalues into Y (row-major ldy: column q).
// The reflectors are applied to all `need` vectors at once, in place in Y's rows: a reflector's
// v is read once for all of them and the loops run over contiguous row segments of Y.
__attribute__((target_clones("avx2,fma", "default")))
bool host_partial_vectors(const HostTridiag& w, int need, double* Y, int ldy) {
  const int m = w.m, lda = w.lda;
  std::vector<double> z((size_t)m * need);
  if (!host_tridiag_eigvectors(w.d.data(), w.e.data(), m, w.theta.data(), need, z.data(),
                               (size_t)m))
    return false;
  for (int r = 0; r < m; ++r)
    for (int q = 0; q < need; ++q) Y[(size_t)r * ldy + q] = z[(size_t)q * m + r];
  std::vector<double> dots(need);
  double* dt = dots.data();
  for (int i = m - 2; i >= 0; --i) {  // Y <- H_i Y, last reflector first
    const double tau = w.tau[i];
    if (tau == 0.0) continue;
    const double* v = w.v.data() + (size_t)i * lda;
    for (int q = 0; q < need; ++q) dt[q] = 0.0;
    for (int k = i + 1; k < m; ++k) {
      const double vk = v[k];
      const double* yr = Y + (size_t)k * ldy;
      for (int q = 0; q < need; ++q) dt[q] = madd(vk, yr[q], dt[q]);
    }
    for (int q = 0; q < need; ++q) dt[q] *= tau;
    for (int k = i + 1; k < m; ++k) {
      const double vk = v[k];
      double* yr = Y + (size_t)k * ldy;
      for (int q = 0; q < need; ++q) yr[q] = madd(-vk, dt[q], yr[q]);
    }
  }
  return true;
}

// host-only export (CPU tests): values (m, descending), vectors (m, need) row-major
extern "C" int sc_host_symmetric_eig_partial(const double* a, int m, int need, double* values,
                                             double* vectors) {
  if (!a || !values || !vectors || m < 1 || need < 1 || need > m) return SC_ERR_INVALID;
  HostTridiag w;
  if (!host_partial_values(a, m, m, &w)) return SC_ERR_NOT_CONVERGED;
  for (int i = 0; i < m; ++i) values[i] = w.theta[i];
  return host_partial_vectors(w, need, vectors, need) ? SC_OK : SC_ERR_NOT_CONVERGED;
}


// ------------------------------------------------------------------------------
// general (non-symmetric) projected eigenproblem, 64 < m <= 128
// ------------------------------------------------------------------------------
// The Rayleigh-Ritz step of the WIDE block Arnoldi (eig_driver.hip: more than 32 eigenpairs of a
// matrix that is not diagonally similar to a symmetric one; reference utils.py:59 calls
// np.linalg.eig on the whole matrix).  Same method as the one-wavefront device solver of
// eig_general.hip, which stops at order 64: real Householder reduction to Hessenberg form,
// explicitly shifted complex QR (Wilkinson shift, exceptional shifts, deflation) to the
// complex Schur form A = Z T Z^H, eigenvectors of T by back substitution, y = Z x.
// Eigenvalues sorted by real part, descending; the first nvec eigenvectors, unit 2-norm, go
// to Y[:, q] = yre + i yim (row-major, ldy).
#include <complex>

namespace {
typedef std::complex<double> cplx;
inline double abs1(const cplx& z) { return std::fabs(z.real()) + std::fabs(z.imag()); }
}  // namespace

bool host_general_eig(const double* a, int lda, int m, int nvec, double* wr, double* wi,
                      double* yre, double* yim, int ldy) {
  if (m <= 0) return true;
  std::vector<double> h((size_t)m * m), z((size_t)m * m, 0.0), v(m);
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < m; ++j) h[(size_t)i * m + j] = a[(size_t)i * lda + j];
    z[(size_t)i * m + i] = 1.0;
  }
  // ---- 1. Hessenberg form, H = P^T A P, Z = P
  for (int k = 0; k + 2 < m; ++k) {
    double below2 = 0.0;
    for (int i = k + 2; i < m; ++i) below2 += h[(size_t)i * m + k] * h[(size_t)i * m + k];
    if (below2 == 0.0) continue;
    const double x0 = h[(size_t)(k + 1) * m + k];
    const double nrm = std::sqrt(below2 + x0 * x0);
    const double alpha = x0 > 0.0 ? -nrm : nrm;
    double vn2 = 0.0;
    for (int i = k + 1; i < m; ++i) {
      v[i] = h[(size_t)i * m + k];
      if (i == k + 1) v[i] -= alpha;
      vn2 += v[i] * v[i];
    }
    const double inv = 1.0 / std::sqrt(vn2);
    for (int i = k + 1; i < m; ++i) v[i] *= inv;
    for (int j = k; j < m; ++j) {  // rows: H <- (I - 2 v v^T) H
      double w = 0.0;
      for (int i = k + 1; i < m; ++i) w += v[i] * h[(size_t)i * m + j];
      w *= 2.0;
      for (int i = k + 1; i < m; ++i) h[(size_t)i * m + j] -= v[i] * w;
    }
    for (int i = 0; i < m; ++i) {  // columns: H <- H (I - 2 v v^T), Z likewise
      double w = 0.0, wz = 0.0;
      for (int j = k + 1; j < m; ++j) {
        w += h[(size_t)i * m + j] * v[j];
        wz += z[(size_t)i * m + j] * v[j];
      }
      w *= 2.0;
      wz *= 2.0;
      for (int j = k + 1; j < m; ++j) {
        h[(size_t)i * m + j] -= w * v[j];
        z[(size_t)i * m + j] -= wz * v[j];
      }
    }
    for (int i = k + 2; i < m; ++i) h[(size_t)i * m + k] = 0.0;
  }
  // ---- 2. complex Schur form by shifted QR
  std::vector<cplx> H((size_t)m * m), Z((size_t)m * m);
  double norm = 0.0;
  for (size_t e = 0; e < (size_t)m * m; ++e) {
    H[e] = h[e];
    Z[e] = z[e];
    norm = std::max(norm, std::fabs(h[e]));
  }
  if (!(norm < 1e300)) return false;  // NaN / inf
  const double eps = 2.220446049250313e-16;
  std::vector<double> cs(m);
  std::vector<cplx> sn(m);
  int en = m - 1, its = 0;
  long budget = 60L * m;
  while (en >= 0) {
    int l = en;
    for (; l > 0; --l) {
      double s = abs1(H[(size_t)(l - 1) * m + l - 1]) + abs1(H[(size_t)l * m + l]);
      if (s == 0.0) s = norm;
      if (abs1(H[(size_t)l * m + l - 1]) <= eps * s) {
        H[(size_t)l * m + l - 1] = 0.0;
        break;
      }
    }
    if (l == en) {
      --en;
      its = 0;
      continue;
    }
    if (--budget < 0 || ++its > 60) return false;
    cplx mu;
    if (its == 10 || its == 20) {  // exceptional shift
      mu = cplx(std::fabs(H[(size_t)en * m + en - 1].real()) +
                    (en >= 2 ? std::fabs(H[(size_t)(en - 1) * m + en - 2].real()) : 0.0),
                0.0) + H[(size_t)en * m + en];
    } else {  // Wilkinson: eigenvalue of the trailing 2 x 2 closer to its last entry
      const cplx a11 = H[(size_t)(en - 1) * m + en - 1], a12 = H[(size_t)(en - 1) * m + en];
      const cplx a21 = H[(size_t)en * m + en - 1], a22 = H[(size_t)en * m + en];
      const cplx half = 0.5 * (a11 - a22);
      const cplx disc = std::sqrt(half * half + a12 * a21);
      const cplx m1 = a22 + half + disc, m2 = a22 + half - disc;  // (a11 + a22) / 2 +- disc
      mu = std::abs(m1 - a22) <= std::abs(m2 - a22) ? m1 : m2;
    }
    for (int i = l; i <= en; ++i) H[(size_t)i * m + i] -= mu;
    for (int k = l; k < en; ++k) {  // H - mu I = Q R: rotations from the left
      const cplx x = H[(size_t)k * m + k], y = H[(size_t)(k + 1) * m + k];
      const double ax = std::abs(x), r = std::hypot(ax, std::abs(y));
      double c;
      cplx s;
      if (r == 0.0) {
        c = 1.0;
        s = 0.0;
      } else if (ax == 0.0) {
        c = 0.0;
        s = std::conj(y) / r;
      } else {
        c = ax / r;
        s = (x / ax) * std::conj(y) / r;
      }
      cs[k] = c;
      sn[k] = s;
      for (int j = k; j < m; ++j) {
        const cplx t0 = H[(size_t)k * m + j], t1 = H[(size_t)(k + 1) * m + j];
        H[(size_t)k * m + j] = c * t0 + s * t1;
        H[(size_t)(k + 1) * m + j] = -std::conj(s) * t0 + c * t1;
      }
      H[(size_t)(k + 1) * m + k] = 0.0;
    }
    for (int k = l; k < en; ++k) {  // R Q: the same rotations from the right (and into Z)
      const double c = cs[k];
      const cplx s = sn[k];
      for (int i = 0; i <= k + 1; ++i) {
        const cplx t0 = H[(size_t)i * m + k], t1 = H[(size_t)i * m + k + 1];
        H[(size_t)i * m + k] = c * t0 + std::conj(s) * t1;
        H[(size_t)i * m + k + 1] = -s * t0 + c * t1;
      }
      for (int i = 0; i < m; ++i) {
        const cplx t0 = Z[(size_t)i * m + k], t1 = Z[(size_t)i * m + k + 1];
        Z[(size_t)i * m + k] = c * t0 + std::conj(s) * t1;
        Z[(size_t)i * m + k + 1] = -s * t0 + c * t1;
      }
    }
    for (int i = l; i <= en; ++i) H[(size_t)i * m + i] += mu;
  }
  // ---- 3. order by real part, eigenvectors of T by back substitution
  std::vector<int> order(m);
  for (int i = 0; i < m; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int p, int q) {
    return H[(size_t)p * m + p].real() > H[(size_t)q * m + q].real();
  });
  for (int q = 0; q < m; ++q) {
    wr[q] = H[(size_t)order[q] * m + order[q]].real();
    wi[q] = H[(size_t)order[q] * m + order[q]].imag();
  }
  std::vector<cplx> x(m), y(m);
  const double small = eps * std::max(norm, 1e-300);
  for (int q = 0; q < std::min(nvec, m); ++q) {
    const int k = order[q];
    const cplx lam = H[(size_t)k * m + k];
    x[k] = 1.0;
    for (int i = k - 1; i >= 0; --i) {
      cplx s = 0.0;
      for (int j = i + 1; j <= k; ++j) s += H[(size_t)i * m + j] * x[j];
      cplx den = H[(size_t)i * m + i] - lam;
      if (abs1(den) < small) den = small;
      x[i] = -s / den;
      if (abs1(x[i]) > 1e150) {  // rescale: the vector is normalised below anyway
        for (int j = i; j <= k; ++j) x[j] *= 1e-150;
      }
    }
    double n2 = 0.0;
    for (int i = 0; i < m; ++i) {
      cplx s = 0.0;
      for (int j = 0; j <= k; ++j) s += Z[(size_t)i * m + j] * x[j];
      y[i] = s;
      n2 += std::norm(s);
    }
    const double inv = n2 > 0.0 ? 1.0 / std::sqrt(n2) : 0.0;
    for (int i = 0; i < m; ++i) {
      yre[(size_t)i * ldy + q] = y[i].real() * inv;
      yim[(size_t)i * ldy + q] = y[i].imag() * inv;
    }
  }
  return true;
}

extern "C" int sc_host_general_eig(const double* a, int m, int nvec, double* values_re,
                                   double* values_im, double* vectors_re, double* vectors_im) {
  if (!a || m <= 0 || nvec < 0 || nvec > m || !values_re || !values_im ||
      (nvec > 0 && (!vectors_re || !vectors_im)))
    return SC_ERR_INVALID;
  return host_general_eig(a, m, m, nvec, values_re, values_im, vectors_re, vectors_im, nvec)
             ? SC_OK
             : SC_ERR_NOT_CONVERGED;
}

// ------------------------------------------------------------------------------
// dense general eigenproblem of order n > 64: the host half
// ------------------------------------------------------------------------------
// np.linalg.eig (reference utils.py:59) on a matrix that is not diagonally similar to a symmetric
// one returns all n eigenpairs.  The device reduces the matrix to upper Hessenberg form
// (eig_general.hip: Householder reflectors, LAPACK dgehd2's storage -- reflector k below the
// subdiagonal of column k, v(k+1) = 1 implied, tau[k]); from there on the work is a serial
// recurrence per eigenvalue, which is why it runs here:
//   eigenvalues   Francis' implicit double-shift QR on the Hessenberg matrix, active block only
//                 (no Schur form is accumulated): the textbook algorithm of EISPACK hqr / LAPACK
//                 dlahqr, incl. the two-consecutive-small-subdiagonals start, the conservative
//                 deflation test and exceptional shifts -- ~10 n^3 flops;
//   eigenvectors  of the few eigenvalues k-means reads: inverse iteration on H - lambda I (LU
//                 with row interchanges of a Hessenberg matrix: one elimination per column,
//                 O(n^2) per solve; LAPACK dhsein / dlaein's method), complex arithmetic for a
//                 complex eigenvalue; close eigenvalues are separated like dhsein does, and the
//                 iterates of a cluster of real eigenvalues are kept independent;
//   x = Q y       through the reflectors (O(n^2) per vector).
// The phase / norm convention of dgeev is applied on the device (k_gen_phase).

bool host_hessenberg_unpack(const double* packed, size_t ld, int n, const double* tau,
                            HostHessenberg* w) {
  w->n = n;
  w->H.assign((size_t)n * n, 0.0);
  w->V.assign((size_t)n * n, 0.0);
  w->tau.assign(tau, tau + std::max(0, n - 2));
  w->norm = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* row = packed + (size_t)i * ld;
    for (int j = std::max(0, i - 1); j < n; ++j) {
      const double v = row[j];
      if (!std::isfinite(v)) return false;
      w->H[(size_t)i * n + j] = v;
      w->norm = std::max(w->norm, std::fabs(v));
    }
    // reflector k (column k, rows k + 2 ..) -> row k of V, contiguous
    for (int k = 0; k + 2 <= i && k + 2 < n; ++k) {
      if (!std::isfinite(row[k])) return false;
      w->V[(size_t)k * n + i] = row[k];
    }
  }
  for (int k = 0; k + 2 < n; ++k) w->V[(size_t)k * n + k + 1] = 1.0;
  return true;
}

namespace {
// LAPACK dlarfg for a vector of 2 or 3 elements (x[0] = alpha): H = I - tau v v^T, v[0] = 1
inline void small_reflector(int nr, double* x, double* tau) {
  double xnorm = nr == 3 ? std::hypot(x[1], x[2]) : std::fabs(x[1]);
  if (xnorm == 0.0) {
    *tau = 0.0;
    return;
  }
  const double alpha = x[0];
  const double beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
  *tau = (beta - alpha) / beta;
  const double scale = 1.0 / (alpha - beta);
  for (int i = 1; i < nr; ++i) x[i] *= scale;
  x[0] = beta;
}
}  // namespace

// Eigenvalues of the upper Hessenberg matrix in w (copied; w is left intact), unordered.
bool host_hessenberg_eigenvalues(const HostHessenberg& w, double* wr, double* wi) {
  const int n = w.n;
  if (n <= 0) return true;
  std::vector<double> hbuf(w.H);
  double* h = hbuf.data();
  auto H = [&](int i, int j) -> double& { return h[(size_t)i * n + j]; };
  const double ulp = 2.220446049250313e-16;
  const double safmin = 2.2250738585072014e-308;
  const double smlnum = safmin * ((double)n / ulp);
  const int itmax = 30 * std::max(10, n);
  int i = n - 1;
  while (i >= 0) {
    int l = 0;
    bool found = false;
    for (int its = 0; its <= itmax; ++its) {
      // ---- a negligible subdiagonal entry splits the active block
      int k;
      for (k = i; k > l; --k) {
        const double sub = std::fabs(H(k, k - 1));
        if (sub <= smlnum) break;
        double tst = std::fabs(H(k - 1, k - 1)) + std::fabs(H(k, k));
        if (tst == 0.0) {
          if (k - 2 >= 0) tst += std::fabs(H(k - 1, k - 2));
          if (k + 1 <= n - 1) tst += std::fabs(H(k + 1, k));
        }
        if (sub <= ulp * tst) {  // (Ahues & Tisseur's conservative criterion)
          const double up = std::fabs(H(k - 1, k));
          const double ab = std::max(sub, up), ba = std::min(sub, up);
          const double dd = std::fabs(H(k - 1, k - 1) - H(k, k)), hk = std::fabs(H(k, k));
          const double aa = std::max(hk, dd), bb = std::min(hk, dd);
          const double s = aa + ab;
          if (ba * (ab / s) <= std::max(smlnum, ulp * (bb * (aa / s)))) break;
        }
      }
      l = k;
      if (l > 0) H(l, l - 1) = 0.0;
      if (l >= i - 1) {
        found = true;
        break;
      }
      // ---- shifts: eigenvalues of the trailing 2 x 2 (exceptional ones now and then)
      double h11, h21, h12, h22;
      if (its > 0 && its % 20 == 0) {
        const double s = std::fabs(H(i, i - 1)) + std::fabs(H(i - 1, i - 2));
        h11 = 0.75 * s + H(i, i);
        h12 = -0.4375 * s;
        h21 = s;
        h22 = h11;
      } else if (its > 0 && its % 10 == 0) {
        const double s = std::fabs(H(l + 1, l)) + std::fabs(H(l + 2, l + 1));
        h11 = 0.75 * s + H(l, l);
        h12 = -0.4375 * s;
        h21 = s;
        h22 = h11;
      } else {
        h11 = H(i - 1, i - 1);
        h21 = H(i, i - 1);
        h12 = H(i - 1, i);
        h22 = H(i, i);
      }
      double rt1r, rt1i, rt2r, rt2i;
      {
        const double s = std::fabs(h11) + std::fabs(h12) + std::fabs(h21) + std::fabs(h22);
        if (s == 0.0) {
          rt1r = rt1i = rt2r = rt2i = 0.0;
        } else {
          h11 /= s;
          h21 /= s;
          h12 /= s;
          h22 /= s;
          const double tr = 0.5 * (h11 + h22);
          const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
          const double rtdisc = std::sqrt(std::fabs(det));
          if (det >= 0.0) {  // complex conjugate shifts
            rt1r = rt2r = tr * s;
            rt1i = rtdisc * s;
            rt2i = -rt1i;
          } else {  // real shifts: the one closer to h22, twice
            rt1r = tr + rtdisc;
            rt2r = tr - rtdisc;
            if (std::fabs(rt1r - h22) <= std::fabs(rt2r - h22)) {
              rt1r *= s;
              rt2r = rt1r;
            } else {
              rt2r *= s;
              rt1r = rt2r;
            }
            rt1i = rt2i = 0.0;
          }
        }
      }
      // ---- start of the bulge: two consecutive small subdiagonal entries
      int m;
      double v[3];
      for (m = i - 2; m >= l; --m) {
        double h21s = std::fabs(H(m + 1, m));
        double s = std::fabs(H(m, m) - rt2r) + std::fabs(rt2i) + h21s;
        h21s = H(m + 1, m) / s;
        v[0] = h21s * H(m, m + 1) + (H(m, m) - rt1r) * ((H(m, m) - rt2r) / s) - rt1i * (rt2i / s);
        v[1] = h21s * (H(m, m) + H(m + 1, m + 1) - rt1r - rt2r);
        v[2] = h21s * H(m + 2, m + 1);
        s = std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]);
        if (s != 0.0) {
          v[0] /= s;
          v[1] /= s;
          v[2] /= s;
        }
        if (m == l) break;
        const double h00 = std::fabs(H(m - 1, m - 1)), hmm = std::fabs(H(m, m)),
                     h11a = std::fabs(H(m + 1, m + 1));
        if (std::fabs(H(m, m - 1)) * (std::fabs(v[1]) + std::fabs(v[2])) <=
            ulp * std::fabs(v[0]) * (h00 + hmm + h11a))
          break;
      }
      // ---- the double-shift sweep on rows / columns l .. i only (no Schur form is kept)
      for (k = m; k <= i - 1; ++k) {
        const int nr = std::min(3, i - k + 1);
        if (k > m) {
          v[0] = H(k, k - 1);
          v[1] = H(k + 1, k - 1);
          v[2] = nr == 3 ? H(k + 2, k - 1) : 0.0;
        }
        double t1;
        small_reflector(nr, v, &t1);
        if (k > m) {
          H(k, k - 1) = v[0];
          H(k + 1, k - 1) = 0.0;
          if (k < i - 1) H(k + 2, k - 1) = 0.0;
        } else if (m > l) {
          H(k, k - 1) *= (1.0 - t1);
        }
        const double v2 = v[1], t2 = t1 * v2;
        if (nr == 3) {
          const double v3 = v[2], t3 = t1 * v3;
          double* r0 = h + (size_t)k * n;
          double* r1 = r0 + n;
          double* r2 = r1 + n;
          for (int j = k; j <= i; ++j) {
            const double sum = r0[j] + v2 * r1[j] + v3 * r2[j];
            r0[j] -= sum * t1;
            r1[j] -= sum * t2;
            r2[j] -= sum * t3;
          }
          const int jend = std::min(k + 3, i);
          for (int j = l; j <= jend; ++j) {
            double* r = h + (size_t)j * n + k;
            const double sum = r[0] + v2 * r[1] + v3 * r[2];
            r[0] -= sum * t1;
            r[1] -= sum * t2;
            r[2] -= sum * t3;
          }
        } else {
          double* r0 = h + (size_t)k * n;
          double* r1 = r0 + n;
          for (int j = k; j <= i; ++j) {
            const double sum = r0[j] + v2 * r1[j];
            r0[j] -= sum * t1;
            r1[j] -= sum * t2;
          }
          for (int j = l; j <= i; ++j) {
            double* r = h + (size_t)j * n + k;
            const double sum = r[0] + v2 * r[1];
            r[0] -= sum * t1;
            r[1] -= sum * t2;
          }
        }
      }
    }
    if (!found) return false;
    if (l == i) {
      wr[i] = H(i, i);
      wi[i] = 0.0;
    } else {  // a 2 x 2 block: both eigenvalues
      const double a = H(i - 1, i - 1), b = H(i - 1, i), c = H(i, i - 1), d = H(i, i);
      const double p = 0.5 * (a - d), bc = b * c, disc = p * p + bc;
      if (disc >= 0.0) {
        const double z = p + std::copysign(std::sqrt(disc), p);
        wr[i - 1] = d + z;
        wr[i] = z != 0.0 ? d - bc / z : d;
        wi[i - 1] = wi[i] = 0.0;
      } else {
        wr[i - 1] = wr[i] = d + p;
        wi[i - 1] = std::sqrt(-disc);
        wi[i] = -wi[i - 1];
      }
    }
    i = l - 1;
  }
  for (int q = 0; q < n; ++q)
    if (!std::isfinite(wr[q]) || !std::isfinite(wi[q])) return false;
  return true;
}

// Eigenvectors of the ORIGINAL matrix (x = Q y, y an eigenvector of H) for `count` of its
// eigenvalues (wr, wi), column-major into vre / vim (column q at + q * ldv); *max_resid: the
// largest ||H y - lambda y||_2 / (||H||_max ||y||_2) seen.  A conjugate partner that follows its
// pair directly is the conjugate vector.  false: an iterate failed to grow.
// One solve is O(n^2) (LU of a Hessenberg matrix + a few substitutions + the back-transform) and
// the solves are independent except inside a cluster of equal real eigenvalues, whose iterates are
// kept independent of each other: the eigenvalues are dealt to up to 16 host threads in groups (a
// real cluster, a conjugate pair, or a single eigenvalue), each thread with its own n x n work
// matrix; real eigenvalues are iterated in real arithmetic.
namespace {
inline double mag1(double z) { return std::fabs(z); }
inline double mag1(const cplx& z) { return abs1(z); }
inline double sq(double z) { return z * z; }
inline double sq(const cplx& z) { return std::norm(z); }
inline double cj(double z) { return z; }
inline cplx cj(const cplx& z) { return std::conj(z); }

// y <- an eigenvector of H for the (already separated) shift lam; prev: earlier vectors of the
// same cluster (real shifts only).  U: n x n scratch.  Returns the relative residual, < 0: failed.
template <typename T>
double hessenberg_inverse_iteration(const HostHessenberg& w, T lam, double eps3,
                                    const std::vector<const std::vector<T>*>& prev,
                                    std::vector<T>* U_store, std::vector<T>* y_out) {
  const int n = w.n;
  const double hnorm = std::max(w.norm, 1e-300);
  std::vector<T>& U = *U_store;
  U.resize((size_t)n * n);
  std::vector<T> mult(n), y(n);
  std::vector<char> swapped(n);
  // ---- LU of H - lambda I with row interchanges: column k eliminates H(k + 1, k)
  for (int i = 0; i < n; ++i) {
    const double* hr = w.H.data() + (size_t)i * n;
    T* ur = U.data() + (size_t)i * n;
    for (int j = 0; j < n; ++j) ur[j] = hr[j];
    ur[i] -= lam;
  }
  for (int k = 0; k + 1 < n; ++k) {
    T* rk = U.data() + (size_t)k * n;
    T* rn = rk + n;
    if (mag1(rk[k]) < mag1(rn[k])) {
      for (int j = k; j < n; ++j) std::swap(rk[j], rn[j]);
      swapped[k] = 1;
    } else {
      swapped[k] = 0;
    }
    if (mag1(rk[k]) == 0.0) rk[k] = eps3;
    const T f = rn[k] / rk[k];
    mult[k] = f;
    if (f != T(0.0))
      for (int j = k + 1; j < n; ++j) rn[j] -= f * rk[j];
    rn[k] = 0.0;
  }
  for (int i = 0; i < n; ++i)
    if (mag1(U[(size_t)i * n + i]) < eps3) U[(size_t)i * n + i] = eps3;  // (dlaein)
  // ---- inverse iteration
  // Gate: an iterate is only accepted with a residual at 1e-8 sqrt(n) ||H|| or below.
  // One that is still above it after six solves gets a different start vector (dlaein's: flat,
  // with one entry pushed the other way); the last attempt of a cluster member gives up the
  // independence from the earlier members -- of a DEFECTIVE eigenvalue there is only one
  // eigenvector, and LAPACK returns it twice as well.  No attempt left: < 0, the caller reports
  // SC_ERR_NOT_CONVERGED instead of handing an unconverged vector to k-means.
  const double rootn = std::sqrt((double)n);
  const double gate = 1e-8 * rootn;
  const int attempts = prev.empty() ? 3 : 4;
  double res = -1.0;
  for (int attempt = 0; attempt < attempts; ++attempt) {
    const bool project = attempt < 3;
    for (int i = 0; i < n; ++i) y[i] = 1.0 / rootn;
    if (attempt > 0 && attempt < 3 && n - attempt >= 0) y[n - attempt] -= rootn / (rootn + 1.0);
    res = -1.0;
    for (int iter = 0; iter < 6; ++iter) {
      if (iter > 0) {  // forward: P and L (the first solve takes its start vector as L^-1 P b)
        for (int k = 0; k + 1 < n; ++k) {
          if (swapped[k]) std::swap(y[k], y[k + 1]);
          y[k + 1] -= mult[k] * y[k];
        }
      }
      for (int i = n - 1; i >= 0; --i) {  // U x = y
        const T* ur = U.data() + (size_t)i * n;
        T s = y[i];
        for (int j = i + 1; j < n; ++j) s -= ur[j] * y[j];
        y[i] = s / ur[i];
      }
      // a cluster of (numerically) equal real eigenvalues: stay independent of its earlier
      // vectors (any basis of the invariant subspace will do; np.linalg.eig returns one)
      if (project) {
        for (const std::vector<T>* pv : prev) {
          T dot = 0.0;
          double nn = 0.0;
          for (int r = 0; r < n; ++r) {
            dot += cj((*pv)[r]) * y[r];
            nn += sq((*pv)[r]);
          }
          if (nn > 0.0)
            for (int r = 0; r < n; ++r) y[r] -= (dot / nn) * (*pv)[r];
        }
      }
      double big = 0.0, n2 = 0.0;
      for (int r = 0; r < n; ++r) big = std::max(big, mag1(y[r]));
      if (!(big > 0.0) || !std::isfinite(big)) {  // this start vector leads nowhere
        res = -1.0;
        break;
      }
      for (int r = 0; r < n; ++r) {
        y[r] /= big;
        n2 += sq(y[r]);
      }
      const double inv = 1.0 / std::sqrt(n2);
      for (int r = 0; r < n; ++r) y[r] *= inv;
      // converged when the residual is at rounding level (one solve with the shift at an
      // eigenvalue already multiplies a generic start vector by ~1 / eps3; two are the rule)
      double res2 = 0.0;
      for (int i = 0; i < n; ++i) {
        const double* hr = w.H.data() + (size_t)i * n;
        T s = -lam * y[i];
        for (int j = std::max(0, i - 1); j < n; ++j) s += hr[j] * y[j];
        res2 += sq(s);
      }
      res = std::sqrt(res2) / hnorm;
      if (iter >= 1 && res <= 1e-12 * rootn) break;
    }
    if (res >= 0.0 && res <= gate) break;
  }
  if (!(res >= 0.0 && res <= gate)) return -1.0;
  *y_out = y;
  return res;
}

// t <- Q t = H_0 H_1 ... H_{n-3} t (the reflectors of the reduction, last first)
template <typename T>
void hessenberg_back_transform(const HostHessenberg& w, std::vector<T>* t_io) {
  const int n = w.n;
  std::vector<T>& t = *t_io;
  for (int k = n - 3; k >= 0; --k) {
    const double tk = w.tau[k];
    if (tk == 0.0) continue;
    const double* v = w.V.data() + (size_t)k * n;
    T dot = 0.0;
    for (int r = k + 1; r < n; ++r) dot += v[r] * t[r];
    dot *= tk;
    for (int r = k + 1; r < n; ++r) t[r] -= dot * v[r];
  }
}
}  // namespace

bool host_hessenberg_vectors(const HostHessenberg& w, const double* wr, const double* wi, int count,
                             double* vre, double* vim, size_t ldv, double* max_resid) {
  const int n = w.n;
  *max_resid = 0.0;
  if (count <= 0) return true;
  const double ulp = 2.220446049250313e-16;
  const double hnorm = std::max(w.norm, 1e-300);
  const double eps3 = hnorm * ulp;
  std::vector<cplx> lam(count);
  for (int q = 0; q < count; ++q) lam[q] = cplx(wr[q], wi[q]);
  // dhsein: an eigenvalue closer than eps3 to one already used is moved by eps3
  for (int q = 0; q < count; ++q) {
    bool again = true;
    while (again) {
      again = false;
      for (int p = 0; p < q; ++p)
        if (abs1(lam[p] - lam[q]) < eps3) {
          lam[q] += eps3;
          again = true;
          break;
        }
    }
  }
  // ---- groups: a conjugate pair | a chain of real eigenvalues closer than cluster_tol | one
  const double cluster_tol = 1e-10 * hnorm;  // real eigenvalues this close: one invariant subspace
  std::vector<std::vector<int>> groups;
  int last_real_group = -1, last_real_q = -1;
  for (int q = 0; q < count; ++q) {
    if (wi[q] != 0.0) {
      if (q > 0 && wi[q] == -wi[q - 1] && wr[q] == wr[q - 1] && !groups.empty() &&
          groups.back().back() == q - 1 && groups.back().size() == 1) {
        groups.back().push_back(q);  // the conjugate of the eigenvalue just before it
      } else {
        groups.push_back({q});
      }
      continue;
    }
    if (last_real_q >= 0 && std::fabs(wr[q] - wr[last_real_q]) <= cluster_tol) {
      groups[last_real_group].push_back(q);
    } else {
      groups.push_back({q});
      last_real_group = (int)groups.size() - 1;
    }
    last_real_q = q;
  }
  std::atomic<int> next{0};
  std::atomic<bool> failed{false};
  std::mutex res_mutex;
  double worst = 0.0;
  auto worker = [&]() {
    std::vector<double> Ur;
    std::vector<cplx> Uc;
    for (;;) {
      const int g = next.fetch_add(1);
      if (g >= (int)groups.size() || failed.load()) return;
      const std::vector<int>& grp = groups[g];
      double local_worst = 0.0;
      if (wi[grp[0]] != 0.0) {  // complex: the vector, and its conjugate for the partner
        std::vector<cplx> y;
        const double res = hessenberg_inverse_iteration<cplx>(w, lam[grp[0]], eps3, {}, &Uc, &y);
        if (res < 0.0) {
          failed.store(true);
          return;
        }
        local_worst = res;
        hessenberg_back_transform(w, &y);
        for (size_t e = 0; e < grp.size(); ++e) {
          const int q = grp[e];
          const double sgn = e == 0 ? 1.0 : -1.0;
          for (int r = 0; r < n; ++r) {
            vre[(size_t)q * ldv + r] = y[r].real();
            vim[(size_t)q * ldv + r] = sgn * y[r].imag();
          }
        }
      } else {  // real eigenvalue(s): real arithmetic, the cluster's members one after the other
        std::vector<std::vector<double>> ys(grp.size());
        for (size_t e = 0; e < grp.size(); ++e) {
          const int q = grp[e];
          std::vector<const std::vector<double>*> prev;
          for (size_t f = 0; f < e; ++f) prev.push_back(&ys[f]);
          const double res =
              hessenberg_inverse_iteration<double>(w, lam[q].real(), eps3, prev, &Ur, &ys[e]);
          if (res < 0.0) {
            failed.store(true);
            return;
          }
          local_worst = std::max(local_worst, res);
          std::vector<double> t(ys[e]);
          hessenberg_back_transform(w, &t);
          for (int r = 0; r < n; ++r) {
            vre[(size_t)q * ldv + r] = t[r];
            vim[(size_t)q * ldv + r] = 0.0;
          }
        }
      }
      std::lock_guard<std::mutex> lock(res_mutex);
      worst = std::max(worst, local_worst);
    }
  };
  const unsigned hw = std::thread::hardware_concurrency();
  // (n^2 work-matrix entries per thread: 16 threads at n = 16384 are 34-69 GB -- fewer there)
  int threads = (int)std::min<size_t>({(size_t)(hw ? hw : 1), (size_t)16, groups.size(),
                                      std::max<size_t>(1, ((size_t)4 << 30) / ((size_t)n * n * 16))});
  if ((size_t)n * n < 65536) threads = 1;  // (small problems: a thread costs more than it saves)
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
  worker();
  for (std::thread& t : pool) t.join();
  *max_resid = worst;
  return !failed.load();
}

// The projected (Rayleigh-Ritz) problems of block Arnoldi, order m <= 128, by the dense route's
// own pieces instead of the complex Schur form above: Householder reduction to Hessenberg form
// (dgehd2, in the storage host_hessenberg_unpack produces), every eigenvalue by the real
// double-shift QR iteration (host_hessenberg_eigenvalues), and only the `nvec` leading
// eigenvectors -- by inverse iteration on the Hessenberg form + back-transform
// (host_hessenberg_vectors).  Same contract as host_general_eig: eigenvalues sorted by real part,
// descending; vectors of unit 2-norm in Y[:, q] (row-major, ldy).  The complex QR with
// accumulated Schur vectors is ~25 m^3 complex operations (4.3 ms at m = 64, 42 ms at m = 128 on a
// host core -- what made the one-wavefront device kernel, 4 ms, worth having); this is ~10 m^3
// real ones + O(nvec m^2): 0.3 / 2 ms.  false: QR or inverse iteration did not converge.
bool host_general_eig_fast(const double* a, int lda, int m, int nvec, double* wr, double* wi,
                           double* yre, double* yim, int ldy) {
  if (m <= 0) return true;
  HostHessenberg hw;
  hw.n = m;
  hw.H.assign((size_t)m * m, 0.0);
  hw.V.assign((size_t)m * m, 0.0);
  hw.tau.assign(std::max(0, m - 2), 0.0);
  std::vector<double>& H = hw.H;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      const double v = a[(size_t)i * lda + j];
      if (!std::isfinite(v)) return false;
      H[(size_t)i * m + j] = v;
    }
  // ---- dgehd2: H <- (I - tau v v^T) H (I - tau v v^T), v = (0 .. 0, 1, v_{k+2} ..)
  std::vector<double> v(m), w(m);
  for (int k = 0; k + 2 < m; ++k) {
    double xnorm2 = 0.0, scale = 0.0;
    for (int i = k + 2; i < m; ++i) scale = std::max(scale, std::fabs(H[(size_t)i * m + k]));
    if (scale == 0.0) continue;  // tau = 0: nothing to annihilate
    for (int i = k + 2; i < m; ++i) {
      const double t = H[(size_t)i * m + k] / scale;
      xnorm2 += t * t;
    }
    const double xnorm = scale * std::sqrt(xnorm2);
    const double alpha = H[(size_t)(k + 1) * m + k];
    const double beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
    const double tau = (beta - alpha) / beta;
    const double inv = 1.0 / (alpha - beta);
    v[k + 1] = 1.0;
    for (int i = k + 2; i < m; ++i) v[i] = H[(size_t)i * m + k] * inv;
    // right: H[:, k+1:] -= tau (H[:, k+1:] v) v^T
    for (int i = 0; i < m; ++i) {
      double dot = 0.0;
      const double* row = H.data() + (size_t)i * m;
      for (int j = k + 1; j < m; ++j) dot += row[j] * v[j];
      w[i] = tau * dot;
    }
    for (int i = 0; i < m; ++i) {
      double* row = H.data() + (size_t)i * m;
      const double wi_ = w[i];
      for (int j = k + 1; j < m; ++j) row[j] -= wi_ * v[j];
    }
    // left: H[k+1:, :] -= tau v (v^T H[k+1:, :])   (column k becomes (beta, 0 ..) by construction)
    for (int j = k + 1; j < m; ++j) w[j] = 0.0;
    for (int i = k + 1; i < m; ++i) {
      const double* row = H.data() + (size_t)i * m;
      const double vi = v[i];
      for (int j = k + 1; j < m; ++j) w[j] += vi * row[j];
    }
    for (int i = k + 1; i < m; ++i) {
      double* row = H.data() + (size_t)i * m;
      const double tv = tau * v[i];
      for (int j = k + 1; j < m; ++j) row[j] -= tv * w[j];
    }
    H[(size_t)(k + 1) * m + k] = beta;
    for (int i = k + 2; i < m; ++i) H[(size_t)i * m + k] = 0.0;
    hw.tau[k] = tau;
    double* vk = hw.V.data() + (size_t)k * m;
    for (int i = k + 1; i < m; ++i) vk[i] = v[i];
  }
  for (int k = 0; k + 2 < m; ++k) hw.V[(size_t)k * m + k + 1] = 1.0;
  hw.norm = 0.0;
  for (size_t e = 0; e < (size_t)m * m; ++e) hw.norm = std::max(hw.norm, std::fabs(H[e]));
  std::vector<double> er(m), ei(m);
  if (!host_hessenberg_eigenvalues(hw, er.data(), ei.data())) return false;
  std::vector<int> order(m);
  for (int i = 0; i < m; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return er[p] > er[q]; });
  for (int i = 0; i < m; ++i) {
    wr[i] = er[order[i]];
    wi[i] = ei[order[i]];
  }
  nvec = std::min(nvec, m);
  if (nvec <= 0) return true;
  std::vector<double> vre((size_t)m * nvec, 0.0), vim((size_t)m * nvec, 0.0);
  double resid = 0.0;
  if (!host_hessenberg_vectors(hw, wr, wi, nvec, vre.data(), vim.data(), (size_t)m, &resid))
    return false;
  for (int q = 0; q < nvec; ++q) {
    double n2 = 0.0;
    for (int r = 0; r < m; ++r)
      n2 += vre[(size_t)q * m + r] * vre[(size_t)q * m + r] + vim[(size_t)q * m + r] * vim[(size_t)q * m + r];
    const double inv = n2 > 0.0 ? 1.0 / std::sqrt(n2) : 0.0;
    for (int r = 0; r < m; ++r) {
      yre[(size_t)r * ldy + q] = vre[(size_t)q * m + r] * inv;
      yim[(size_t)r * ldy + q] = vim[(size_t)q * m + r] * inv;
    }
  }
  return true;
}

extern "C" int sc_host_general_eig_fast(const double* a, int m, int nvec, double* values_re,
                                        double* values_im, double* vectors_re,
                                        double* vectors_im) {
  if (!a || m <= 0 || nvec < 0 || nvec > m || !values_re || !values_im ||
      (nvec > 0 && (!vectors_re || !vectors_im)))
    return SC_ERR_INVALID;
  return host_general_eig_fast(a, m, m, nvec, values_re, values_im, vectors_re, vectors_im, nvec)
             ? SC_OK
             : SC_ERR_NOT_CONVERGED;
}

// host-only exports (CPU tests): `packed` (n, n) row-major in the device reduction's storage
// (Hessenberg matrix on and above the subdiagonal, reflectors below it), tau (n - 2).
extern "C" int sc_host_hessenberg_eig(const double* packed, const double* tau, int n, int count,
                                      const int32_t* pick, double* values_re, double* values_im,
                                      double* vectors_re, double* vectors_im, double* max_resid) {
  if (!packed || n < 1 || count < 0 || count > n || !values_re || !values_im ||
      (n > 2 && !tau) || (count > 0 && (!pick || !vectors_re || !vectors_im)))
    return SC_ERR_INVALID;
  HostHessenberg w;
  if (!host_hessenberg_unpack(packed, (size_t)n, n, tau, &w)) return SC_ERR_NON_FINITE;
  if (!host_hessenberg_eigenvalues(w, values_re, values_im)) return SC_ERR_NOT_CONVERGED;
  if (count == 0) return SC_OK;
  std::vector<double> pr(count), pi(count), cre((size_t)n * count), cim((size_t)n * count);
  for (int q = 0; q < count; ++q) {
    if (pick[q] < 0 || pick[q] >= n) return SC_ERR_INVALID;
    pr[q] = values_re[pick[q]];
    pi[q] = values_im[pick[q]];
  }
  double res = 0.0;
  if (!host_hessenberg_vectors(w, pr.data(), pi.data(), count, cre.data(), cim.data(), (size_t)n,
                               &res))
    return SC_ERR_NOT_CONVERGED;
  if (max_resid) *max_resid = res;
  for (int q = 0; q < count; ++q)
    for (int r = 0; r < n; ++r) {  // (n, count) row-major out
      vectors_re[(size_t)r * count + q] = cre[(size_t)q * n + r];
      vectors_im[(size_t)r * count + q] = cim[(size_t)q * n + r];
    }
  return SC_OK;
}
