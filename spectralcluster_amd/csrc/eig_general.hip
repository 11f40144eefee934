// General (non-symmetric) eigenproblem of the path:
// np.linalg.eig + .real + argsort of reference utils.py:44-71 for matrices that are NOT
// diagonally similar to a symmetric one (a refinement sequence ending in RowWiseThreshold,
// a non-symmetric constraint matrix, ...).
//
// Small dense solver (order m <= 64), ONE wavefront, everything in LDS:
//   1. Householder reduction to upper Hessenberg form (real arithmetic),
//   2. explicitly shifted complex QR iteration (Wilkinson shift, exceptional shifts at
//      iterations 10 / 20, deflation on negligible subdiagonals) -> complex Schur form
//      A = Z T Z^H with the Schur vectors accumulated,
//   3. eigenvectors of T by back substitution, y = Z x, unit 2-norm,
//   4. eigenvalues sorted by real part, descending.
// Lane j owns column j in the left (row-mixing) half of a QR sweep and row j in the right
// (column-mixing) half, so a sweep needs no cross-lane traffic except the 2x2 rotation that
// lane k broadcasts at step k (wave shuffles).  The value carried from step k to k + 1 stays
// in a register; LDS is touched once per step.
//
// Large problems use it as the Rayleigh-Ritz step of a block Arnoldi iteration (eig_driver.hip);
// the tall-skinny helpers for that (explicit residuals, Ritz vectors, LAPACK's phase
// normalisation of complex eigenvectors) are below.
#include <hip/hip_runtime.h>

#include "sc_internal.h"

namespace sc {

constexpr int GM = kGenMax;   // 64: one lane per row / column
constexpr int GLD = GM + 1;   // odd stride: row walks do not pile on one LDS bank
static_assert(GM == 64, "k_gen_eig maps one lane to one row/column of the matrix");

struct cd {
  double re, im;
};
__device__ __forceinline__ cd cmul(cd a, cd b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cd cmulc(cd a, cd b) {  // a * conj(b)
  return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im};
}
__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cd cscale(cd a, double s) { return {a.re * s, a.im * s}; }
__device__ __forceinline__ cd cdiv(cd a, cd b) {
  // Smith's algorithm: no overflow in the squares
  if (fabs(b.re) >= fabs(b.im)) {
    const double t = b.im / b.re, den = b.re + b.im * t;
    return {(a.re + a.im * t) / den, (a.im - a.re * t) / den};
  }
  const double t = b.re / b.im, den = b.re * t + b.im;
  return {(a.re * t + a.im) / den, (a.im * t - a.re) / den};
}
__device__ __forceinline__ cd csqrt_(cd z) {
  const double r = hypot(z.re, z.im);
  const double t = sqrt(0.5 * (r + fabs(z.re)));
  if (t == 0.0) return {0.0, 0.0};
  if (z.re >= 0.0) return {t, z.im / (2.0 * t)};
  return {fabs(z.im) / (2.0 * t), copysign(t, z.im)};
}
__device__ __forceinline__ cd shfl_cd(cd v, int src) {
  return {__shfl(v.re, src), __shfl(v.im, src)};
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// info[0] = 0 ok / 1 QR iteration did not converge; info[1] = sweeps used
__global__ __launch_bounds__(64) void k_gen_eig(const double* __restrict__ A, int lda, int m,
                                                double sign, int nvec,
                                                double* __restrict__ theta_re,
                                                double* __restrict__ theta_im,
                                                double* __restrict__ Yre,
                                                double* __restrict__ Yim, int ldy,
                                                int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Hre = smem;
  double* Him = Hre + GM * GLD;
  double* Zre = Him + GM * GLD;
  double* Zim = Zre + GM * GLD;
  double* rot = Zim + GM * GLD;  // [4][GM]: c.re, c.im, s.re, s.im of the sweep
  double* xs = rot + 4 * GM;     // [2][GM]: Householder vector / eigenvector of T
  int* order = reinterpret_cast<int*>(xs + 2 * GM);  // [GM] Schur index of rank q
  const int lane = threadIdx.x;
  const bool in = lane < m;
#define HR(i, j) Hre[(i) * GLD + (j)]
#define HI(i, j) Him[(i) * GLD + (j)]
#define ZR(i, j) Zre[(i) * GLD + (j)]
#define ZI(i, j) Zim[(i) * GLD + (j)]
  for (int i = 0; i < m; ++i) {
    if (in) {
      HR(i, lane) = sign * A[(size_t)i * lda + lane];
      HI(i, lane) = 0.0;
      ZR(i, lane) = i == lane ? 1.0 : 0.0;
      ZI(i, lane) = 0.0;
    }
  }
  __syncthreads();

  // ---- 1. Householder reduction to Hessenberg form (real) -----------------------
  for (int k = 0; k + 2 < m; ++k) {
    const double x = (lane > k && in) ? HR(lane, k) : 0.0;
    const double nrm2 = wave_sum(x * x);
    const double below2 = wave_sum(lane > k + 1 ? x * x : 0.0);
    if (below2 == 0.0) continue;  // column already reduced (uniform branch)
    const double xk1 = __shfl(x, k + 1);
    const double alpha = xk1 > 0.0 ? -sqrt(nrm2) : sqrt(nrm2);
    double v = x;
    if (lane == k + 1) v -= alpha;
    const double vn2 = wave_sum(v * v);
    v = v / sqrt(vn2);
    xs[lane] = v;
    __syncthreads();
    if (lane >= k && in) {  // H <- (I - 2 v v^T) H : lane = column
      double w = 0.0;
      for (int i = k + 1; i < m; ++i) w += xs[i] * HR(i, lane);
      w *= 2.0;
      for (int i = k + 1; i < m; ++i) HR(i, lane) -= xs[i] * w;
    }
    __syncthreads();
    if (in) {  // H <- H (I - 2 v v^T), Z likewise : lane = row
      double u = 0.0, uz = 0.0;
      for (int j = k + 1; j < m; ++j) {
        u += HR(lane, j) * xs[j];
        uz += ZR(lane, j) * xs[j];
      }
      u *= 2.0;
      uz *= 2.0;
      for (int j = k + 1; j < m; ++j) {
        HR(lane, j) -= u * xs[j];
        ZR(lane, j) -= uz * xs[j];
      }
    }
    __syncthreads();
  }
  if (in)
    for (int i = lane + 2; i < m; ++i) HR(i, lane) = 0.0;  // exact zeros below the subdiagonal
  __syncthreads();

  // ---- 2. shifted complex QR iteration -> Schur form ----------------------------
  double norm1 = 0.0;
  if (in)
    for (int j = 0; j < m; ++j) norm1 += fabs(HR(lane, j));
  norm1 = wave_sum(norm1);
  const double eps = 2.220446049250313e-16;
  int en = m - 1, its = 0, sweeps = 0;
  bool failed = false;
  while (en >= 0) {
    bool small = false;
    if (lane >= 1 && lane <= en) {
      const double sd = fabs(HR(lane, lane - 1)) + fabs(HI(lane, lane - 1));
      double sc = fabs(HR(lane - 1, lane - 1)) + fabs(HI(lane - 1, lane - 1)) +
                  fabs(HR(lane, lane)) + fabs(HI(lane, lane));
      if (sc == 0.0) sc = norm1;
      small = sd <= eps * sc;
    }
    const unsigned long long mask = __ballot(small);
    const int l = mask ? 63 - __clzll((long long)mask) : 0;  // largest negligible subdiagonal
    if (l > 0 && lane == l) {
      HR(l, l - 1) = 0.0;
      HI(l, l - 1) = 0.0;
    }
    if (l == en) {  // eigenvalue en has converged
      --en;
      its = 0;
      __syncthreads();
      continue;
    }
    if (its >= 60) {
      failed = true;
      break;
    }
    cd sh;
    if (its == 10 || its == 20) {
      sh = {fabs(HR(en, en - 1)) + (en >= 2 ? fabs(HR(en - 1, en - 2)) : 0.0), 0.0};
    } else {  // Wilkinson: eigenvalue of the trailing 2x2 closer to H[en][en]
      const cd a = {HR(en - 1, en - 1), HI(en - 1, en - 1)};
      const cd b = {HR(en - 1, en), HI(en - 1, en)};
      const cd c = {HR(en, en - 1), HI(en, en - 1)};
      const cd d = {HR(en, en), HI(en, en)};
      const cd half = cscale(cadd(a, d), 0.5);
      // ((a - d) / 2)^2 + b c, not (tr / 2)^2 - det: for a nearly defective 2x2 (a ~ d,
      // b c tiny) the latter cancels catastrophically and the iteration stagnates
      const cd hd = cscale(csub(a, d), 0.5);
      const cd disc = csqrt_(cadd(cmul(hd, hd), cmul(b, c)));
      const cd l1 = cadd(half, disc), l2 = csub(half, disc);
      const cd d1 = csub(l1, d), d2 = csub(l2, d);
      sh = (d1.re * d1.re + d1.im * d1.im) < (d2.re * d2.re + d2.im * d2.im) ? l1 : l2;
    }
    ++its;
    ++sweeps;
    __syncthreads();
    if (lane >= l && lane <= en) {
      HR(lane, lane) -= sh.re;
      HI(lane, lane) -= sh.im;
    }
    __syncthreads();
    // -- left half: R = G_{en-1} ... G_l (H - sh I); lane = column, rows l..en
    {
      cd carry = {0.0, 0.0};  // current value of H[k][lane]
      if (lane >= l && in) carry = {HR(l, lane), HI(l, lane)};
      cd below = {0.0, 0.0};
      if (lane >= l && in) below = {HR(l + 1, lane), HI(l + 1, lane)};
      for (int k = l; k < en; ++k) {
        cd nxt = {0.0, 0.0};  // prefetch row k + 2 for the next step
        if (k + 2 <= en && lane >= k + 1 && in) nxt = {HR(k + 2, lane), HI(k + 2, lane)};
        const cd x = shfl_cd(carry, k), y = shfl_cd(below, k);
        const double r = sqrt(x.re * x.re + x.im * x.im + y.re * y.re + y.im * y.im);
        cd c = {1.0, 0.0}, s = {0.0, 0.0};
        if (r > 0.0) {
          c = cscale(x, 1.0 / r);
          s = cscale(y, 1.0 / r);
        }
        if (lane == 0) {
          rot[k] = c.re;
          rot[GM + k] = c.im;
          rot[2 * GM + k] = s.re;
          rot[3 * GM + k] = s.im;
        }
        if (lane >= k && in) {
          // [row k; row k+1] <- [conj(c) conj(s); -s c] [row k; row k+1]
          const cd nk = cadd(cmulc(carry, c), cmulc(below, s));
          const cd nk1 = csub(cmul(c, below), cmul(s, carry));
          HR(k, lane) = nk.re;
          HI(k, lane) = nk.im;
          carry = nk1;
          if (lane == k) {
            HR(k + 1, k) = 0.0;
            HI(k + 1, k) = 0.0;
          }
        }
        below = nxt;
      }
      if (lane >= en && in) {
        HR(en, lane) = carry.re;
        HI(en, lane) = carry.im;
      }
    }
    __syncthreads();
    // -- right half: H = R G_l^H ... G_{en-1}^H + sh I, Z <- Z G^H; lane = row
    {
      cd hc = {0.0, 0.0}, zc = {0.0, 0.0};
      if (in) zc = {ZR(lane, l), ZI(lane, l)};
      for (int k = l; k < en; ++k) {
        const cd c = {rot[k], rot[GM + k]}, s = {rot[2 * GM + k], rot[3 * GM + k]};
        // rows of R with a nonzero in column k or k+1: i <= k + 1
        const bool act = lane <= k + 1 && lane <= en;
        if (act) {
          if (k == l || lane == k + 1) hc = {HR(lane, k), HI(lane, k)};
          const cd nx = {HR(lane, k + 1), HI(lane, k + 1)};
          const cd nk = cadd(cmul(hc, c), cmul(nx, s));
          const cd nk1 = csub(cmulc(nx, c), cmulc(hc, s));
          HR(lane, k) = nk.re;
          HI(lane, k) = nk.im;
          hc = nk1;
        }
        if (in) {
          const cd nx = {ZR(lane, k + 1), ZI(lane, k + 1)};
          const cd nk = cadd(cmul(zc, c), cmul(nx, s));
          const cd nk1 = csub(cmulc(nx, c), cmulc(zc, s));
          ZR(lane, k) = nk.re;
          ZI(lane, k) = nk.im;
          zc = nk1;
        }
      }
      if (lane <= en) {
        HR(lane, en) = hc.re;
        HI(lane, en) = hc.im;
      }
      if (in) {
        ZR(lane, en) = zc.re;
        ZI(lane, en) = zc.im;
      }
    }
    __syncthreads();
    if (lane >= l && lane <= en) {
      HR(lane, lane) += sh.re;
      HI(lane, lane) += sh.im;
    }
    __syncthreads();
  }
  if (lane == 0) {
    info[0] = failed ? 1 : 0;
    info[1] = sweeps;
  }
  if (failed) return;

  // ---- 3. order by real part (descending; ties by Schur index) ------------------
  const double my_re = in ? HR(lane, lane) : 0.0;
  const double my_im = in ? HI(lane, lane) : 0.0;
  xs[lane] = my_re;
  __syncthreads();
  if (in) {
    int rank = 0;
    for (int j = 0; j < m; ++j) rank += (xs[j] > my_re) || (xs[j] == my_re && j < lane);
    order[rank] = lane;
    theta_re[rank] = my_re;
    theta_im[rank] = my_im;
  }
  double tnorm = 0.0;
  if (in)
    for (int j = lane; j < m; ++j) tnorm += fabs(HR(lane, j)) + fabs(HI(lane, j));
  tnorm = wave_sum(tnorm);
  const double smin = fmax(eps * tnorm, 2.2250738585072014e-308 * m / eps);
  __syncthreads();

  // ---- 4. eigenvectors: (T - lambda_k I) x = 0, y = Z x --------------------------
  for (int q = 0; q < nvec && q < m; ++q) {
    const int k = order[q];
    const cd lk = {HR(k, k), HI(k, k)};
    cd r = {0.0, 0.0}, x = {0.0, 0.0};
    if (lane < k) r = {-HR(lane, k), -HI(lane, k)};
    if (lane == k) x = {1.0, 0.0};
    for (int j = k - 1; j >= 0; --j) {
      cd d = {HR(j, j) - lk.re, HI(j, j) - lk.im};
      if (fabs(d.re) + fabs(d.im) < smin) d = {smin, 0.0};
      const cd xj = cdiv(shfl_cd(r, j), d);
      if (lane == j) x = xj;
      if (lane < j) r = csub(r, cmul({HR(lane, j), HI(lane, j)}, xj));
    }
    __syncthreads();  // previous iteration's reads of xs are done
    xs[lane] = x.re;
    xs[GM + lane] = x.im;
    __syncthreads();
    cd y = {0.0, 0.0};
    if (in)
      for (int j = 0; j <= k; ++j)
        y = cadd(y, cmul({ZR(lane, j), ZI(lane, j)}, {xs[j], xs[GM + j]}));
    const double nn = wave_sum(y.re * y.re + y.im * y.im);
    const double inv = nn > 0.0 ? 1.0 / sqrt(nn) : 0.0;
    if (in) {
      Yre[(size_t)lane * ldy + q] = y.re * inv;
      Yim[(size_t)lane * ldy + q] = y.im * inv;
    }
  }
#undef HR
#undef HI
#undef ZR
#undef ZI
}

// ---- explicit residuals || Op Q y - theta Q y ||_2 of `cols` Ritz pairs ------------
// grid: row blocks of 8; block: 8 rows x 32 columns.  partial[blk * 32 + c] = sum of
// |res|^2 over the block's rows; k_gen_residual_reduce finishes (fixed order).
__global__ __launch_bounds__(256) void k_gen_residual(
    const double* __restrict__ Q, const double* __restrict__ OpQ, int ldq, int m, int n,
    const double* __restrict__ Yre, const double* __restrict__ Yim, int ldy,
    const double* __restrict__ theta_re, const double* __restrict__ theta_im, int cols,
    double* __restrict__ partial) {
  // Ritz coefficients of the launch's 32 pairs: m x 33 doubles each (m <= 64 on the narrow
  // Arnoldi, up to kLdq on the wide one: dynamic LDS, 2 * m * 33 * 8 bytes)
  extern __shared__ __attribute__((aligned(16))) double ysm[];
  double (*yr)[33] = reinterpret_cast<double (*)[33]>(ysm);
  double (*yi)[33] = yr + m;
  __shared__ double red[8][33];
  const int c = threadIdx.x & 31, rr = threadIdx.x >> 5;
  for (int e = threadIdx.x; e < m * 32; e += 256) {
    const int j = e >> 5, cc = e & 31;
    yr[j][cc] = cc < cols ? Yre[(size_t)j * ldy + cc] : 0.0;
    yi[j][cc] = cc < cols ? Yim[(size_t)j * ldy + cc] : 0.0;
  }
  __syncthreads();
  const int row = blockIdx.x * 8 + rr;
  double acc = 0.0;
  if (row < n && c < cols) {
    double qr = 0.0, qi = 0.0, orr = 0.0, oi = 0.0;
    for (int j = 0; j < m; ++j) {
      const double qv = Q[(size_t)row * ldq + j], ov = OpQ[(size_t)row * ldq + j];
      qr += qv * yr[j][c];
      qi += qv * yi[j][c];
      orr += ov * yr[j][c];
      oi += ov * yi[j][c];
    }
    const double tr = theta_re[c], ti = theta_im[c];
    const double rre = orr - (tr * qr - ti * qi);
    const double rim = oi - (tr * qi + ti * qr);
    acc = rre * rre + rim * rim;
  }
  red[rr][c] = acc;
  __syncthreads();
  if (rr == 0) {
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += red[r][c];
    partial[(size_t)blockIdx.x * 32 + c] = sum;
  }
}
__global__ void k_gen_residual_reduce(const double* __restrict__ partial, int nblocks,
                                      int cols, double* __restrict__ resid) {
  const int c = threadIdx.x;
  if (c >= cols) return;
  double sum = 0.0;
  for (int b = 0; b < nblocks; ++b) sum += partial[(size_t)b * 32 + c];
  resid[c] = sqrt(sum);
}

// ---- Ritz vectors V = Q Y (complex), column-major: V[c * ldv + row] ------------------
// Q == nullptr: the basis is the identity (dense path), V = Y.
__global__ __launch_bounds__(256) void k_gen_ritz(const double* __restrict__ Q, int ldq, int m,
                                                 int n, const double* __restrict__ Yre,
                                                 const double* __restrict__ Yim, int ldy,
                                                 int col0, int cols, double* __restrict__ Vre,
                                                 double* __restrict__ Vim, int ldv) {
  extern __shared__ __attribute__((aligned(16))) double ysm[];
  double (*yr)[33] = reinterpret_cast<double (*)[33]>(ysm);
  double (*yi)[33] = yr + m;
  const int c = threadIdx.x & 31, rr = threadIdx.x >> 5;
  for (int e = threadIdx.x; e < m * 32; e += 256) {
    const int j = e >> 5, cc = e & 31;
    yr[j][cc] = cc < cols ? Yre[(size_t)j * ldy + col0 + cc] : 0.0;
    yi[j][cc] = cc < cols ? Yim[(size_t)j * ldy + col0 + cc] : 0.0;
  }
  __syncthreads();
  const int row = blockIdx.x * 8 + rr;
  if (row >= n || c >= cols) return;
  double vr = 0.0, vi = 0.0;
  if (Q == nullptr) {
    vr = yr[row][c];
    vi = yi[row][c];
  } else {
    for (int j = 0; j < m; ++j) {
      const double qv = Q[(size_t)row * ldq + j];
      vr += qv * yr[j][c];
      vi += qv * yi[j][c];
    }
  }
  Vre[(size_t)(col0 + c) * ldv + row] = vr;
  Vim[(size_t)(col0 + c) * ldv + row] = vi;
}

// ---- LAPACK's eigenvector normalisation (dgeev): unit 2-norm, and for a complex vector
// the component of largest magnitude (first one on ties) rotated onto the real axis.
// One workgroup per column; rewrites V in place and stores the real part into E.
__global__ __launch_bounds__(256) void k_gen_phase(double* __restrict__ Vre,
                                                  double* __restrict__ Vim, int ldv, int n,
                                                  double* __restrict__ E, int lde) {
  __shared__ double s_sum[256], s_max[256];
  __shared__ int s_idx[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double* vr = Vre + (size_t)c * ldv;
  double* vi = Vim + (size_t)c * ldv;
  double sum = 0.0, best = -1.0;
  int besti = 0;
  for (int r = tid; r < n; r += 256) {
    const double mag = vr[r] * vr[r] + vi[r] * vi[r];
    sum += mag;
    if (mag > best) {
      best = mag;
      besti = r;
    }
  }
  s_sum[tid] = sum;
  s_max[tid] = best;
  s_idx[tid] = besti;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      s_sum[tid] += s_sum[tid + o];
      if (s_max[tid + o] > s_max[tid] ||
          (s_max[tid + o] == s_max[tid] && s_idx[tid + o] < s_idx[tid])) {
        s_max[tid] = s_max[tid + o];
        s_idx[tid] = s_idx[tid + o];
      }
    }
    __syncthreads();
  }
  const double norm = sqrt(s_sum[0]);
  const int k = s_idx[0];
  const double inv = norm > 0.0 ? 1.0 / norm : 0.0;
  const double pr = vr[k], pi = vi[k];
  const double pm = sqrt(pr * pr + pi * pi);
  // multiply by conj(v_k) / |v_k| (a vector that is real up to rounding stays as it is)
  const double cr = pm > 0.0 ? pr / pm : 1.0, ci = pm > 0.0 ? -pi / pm : 0.0;
  __syncthreads();
  for (int r = tid; r < n; r += 256) {
    const double a = vr[r] * inv, b = vi[r] * inv;
    const double re = a * cr - b * ci, im = a * ci + b * cr;
    vr[r] = re;
    vi[r] = im;
    if (E != nullptr) E[(size_t)c * lde + r] = re;
  }
}

// ---- restart block: W[row * B + j] = (part[j] ? Vim : Vre)[src[j]][row], or noise ----
__global__ void k_gen_gather(const double* __restrict__ Vre, const double* __restrict__ Vim,
                             int ldv, int n, const int* __restrict__ src, uint64_t seed,
                             double* __restrict__ W) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * kEigBlock) return;
  const int row = e / kEigBlock, j = e % kEigBlock;
  const int code = src[j];  // < 0: random column; else 2 * col + part
  double v;
  if (code < 0) {
    uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(e + 1));
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    v = (double)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  } else {
    v = ((code & 1) ? Vim : Vre)[(size_t)(code >> 1) * ldv + row];
  }
  W[e] = v;
}

// ---- operator scalings of the general path:  Op x = p .* x + cl .* (M (cr .* x)) ------
//   None / Affinity : Op =  M                     cl = cr = 1,        p = 0
//   Unnormalized    : Op = -(D - M)               cl = cr = 1,        p = -deg
//   RandomWalk      : Op = -D'^-1 (D - M)         cl = g, cr = 1,     p = -g deg,  g = 1/(deg+eps)
//   GraphCut        : Op = -D'^-1/2 (D-M) D'^-1/2 cl = cr = h,        p = -h^2 deg, h = 1/(sqrt(deg)+eps)
__global__ void k_scaling_general(const double* __restrict__ deg, int n, int laplacian_type,
                                  double* __restrict__ cl, double* __restrict__ cr,
                                  double* __restrict__ p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = deg[i];
  double l = 1.0, r = 1.0, pp = 0.0;
  if (laplacian_type == SC_LAPLACIAN_UNNORMALIZED) {
    pp = -d;
  } else if (laplacian_type == SC_LAPLACIAN_RANDOM_WALK) {
    l = 1.0 / (d + 1e-10);
    pp = -(l * d);
  } else if (laplacian_type == SC_LAPLACIAN_GRAPH_CUT) {
    l = r = 1.0 / (sqrt(d) + 1e-10);
    pp = -((l * d) * r);
  }
  cl[i] = l;
  cr[i] = r;
  p[i] = pp;
}

// -------------------------------------------------------------------------------------
void launch_gen_eig(hipStream_t s, const double* A, int lda, int m, double sign, int nvec,
                    double* theta_re, double* theta_im, double* Yre, double* Yim, int ldy,
                    int* info) {
  const size_t lds = sizeof(double) * (4 * (size_t)GM * GLD + 6 * GM) + sizeof(int) * GM + 64;
  SC_OPT_IN_LDS(k_gen_eig, 160 * 1024 - 256);
  hipLaunchKernelGGL(k_gen_eig, dim3(1), dim3(64), lds, s, A, lda, m, sign, nvec, theta_re,
                     theta_im, Yre, Yim, ldy, info);
}
int gen_residual_blocks(int n) { return (n + 7) / 8; }
void launch_gen_residual(hipStream_t s, const double* Q, const double* OpQ, int ldq, int m,
                         int n, const double* Yre, const double* Yim, int ldy,
                         const double* theta_re, const double* theta_im, int cols,
                         double* partial, double* resid) {
  const int nb = gen_residual_blocks(n);
  const size_t lds = (size_t)2 * m * 33 * sizeof(double);
  SC_OPT_IN_LDS(k_gen_residual, 2 * kLdq * 33 * (int)sizeof(double));
  hipLaunchKernelGGL(k_gen_residual, dim3(nb), dim3(256), lds, s, Q, OpQ, ldq, m, n, Yre, Yim,
                     ldy, theta_re, theta_im, cols, partial);
  hipLaunchKernelGGL(k_gen_residual_reduce, dim3(1), dim3(32), 0, s, partial, nb, cols, resid);
}
void launch_gen_ritz(hipStream_t s, const double* Q, int ldq, int m, int n, const double* Yre,
                     const double* Yim, int ldy, int cols, double* Vre, double* Vim, int ldv) {
  const size_t lds = (size_t)2 * m * 33 * sizeof(double);
  SC_OPT_IN_LDS(k_gen_ritz, 2 * kLdq * 33 * (int)sizeof(double));
  for (int c0 = 0; c0 < cols; c0 += 32)
    hipLaunchKernelGGL(k_gen_ritz, dim3((n + 7) / 8), dim3(256), lds, s, Q, ldq, m, n, Yre, Yim,
                       ldy, c0, cols - c0 < 32 ? cols - c0 : 32, Vre, Vim, ldv);
}
void launch_gen_phase(hipStream_t s, double* Vre, double* Vim, int ldv, int n, int cols,
                      double* E, int lde) {
  hipLaunchKernelGGL(k_gen_phase, dim3(cols), dim3(256), 0, s, Vre, Vim, ldv, n, E, lde);
}
void launch_gen_gather(hipStream_t s, const double* Vre, const double* Vim, int ldv, int n,
                       const int* src, uint64_t seed, double* W) {
  hipLaunchKernelGGL(k_gen_gather, dim3((n * kEigBlock + 255) / 256), dim3(256), 0, s, Vre,
                     Vim, ldv, n, src, seed, W);
}
__global__ void k_negate2(const double* __restrict__ a, const double* __restrict__ b, int n,
                          double* __restrict__ out, size_t stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = -a[i];
  out[stride + i] = -b[i];
}
void launch_negate2(hipStream_t s, const double* a, const double* b, int n, double* out,
                    size_t stride) {
  hipLaunchKernelGGL(k_negate2, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, out, stride);
}
void launch_scaling_general(hipStream_t s, const double* deg, int n, int laplacian_type,
                            double* cl, double* cr, double* p) {
  hipLaunchKernelGGL(k_scaling_general, dim3((n + 255) / 256), dim3(256), 0, s, deg, n,
                     laplacian_type, cl, cr, p);
}

}  // namespace sc
