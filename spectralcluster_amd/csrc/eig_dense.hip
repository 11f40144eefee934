// Dense symmetric eigen path (reference utils.py:44-71).  Two users:
//   * every eigenvalue is consumed: max_clusters=None with a Laplacian reads w[1..n-1]
//     (utils.py:100-115), the ascending NormalizedDiff gap also reads np.max(w) (utils.py:110),
//     a max_clusters above what a Krylov basis holds;
//   * the landing pad of spectra block Lanczos gives up on (eig_driver.hip: dense_fallback).
//
//   Op = diag(p) + diag(c) S diag(c)  -- materialised once into a scratch matrix,
//   blocked Householder tridiagonalisation  T = Q^T Op Q  (LAPACK dsytrd / dlatrd, below),
//   every eigenvalue of T by Sturm-sequence multisection (LAPACK dstebz's count).
//
// Why Sturm counts and not implicit QL on the tridiagonal: QL is one serial chain of ~3 n^2
// rotations; the count is independent per eigenvalue and per shift -- the same eigenvalues to
// ulp * ||T||, in parallel.
// Eigenvectors: the few columns k-means takes normally come from the block Lanczos solver
// (eig.hip) on the untouched S.  When that solver cannot deliver them (clustered spectra:
// no convergence within its restart budget) they come from HERE, like LAPACK dstein +
// dormtr: inverse iteration on the tridiagonal form (host, O(n) per vector and iteration:
// host_eig.cpp) and the back-transform Q z on the device (k_td_backtransform).  For that the
// reflectors are kept: v_j (v_j[j+1] = 1) in ROW j of the destroyed matrix, columns j+1 ..
// n-1 -- row j is dead once column j is eliminated and a row is contiguous, so the
// back-transform streams whole rows -- and tau_j in taus[j].
#include <algorithm>

#include "sc_internal.h"

namespace sc {

// ---------------------------------------------------------------- materialise Op
__global__ __launch_bounds__(256) void k_td_materialize(
    const double* __restrict__ S, int ld, int n, const double* __restrict__ c,
    const double* __restrict__ p, double* __restrict__ M) {
  const int i = blockIdx.y;
  const double ci = c[i], pi = p[i];
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    double v = (ci * c[j]) * S[(size_t)i * ld + j];  // (c_i c_j) first: exactly symmetric
    if (i == j) v += pi;
    M[(size_t)i * ld + j] = v;
  }
}

// ---------------------------------------------------------------- block reductions
__device__ __forceinline__ double td_block_sum(double v, double* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += sm[w];  // fixed order, same in every thread
  return s;
}

// ---------------------------------------------------------------- blocked tridiagonalisation
// LAPACK dsytrd / dlatrd (panels of kTdNb columns) with the reductions deferred to launch
// boundaries, TWO multi-workgroup launches per column and no single-workgroup kernel on the
// critical path (round 2's unblocked dsytd2 form: 2 launches per column too, but one of them
// a single workgroup walking 2 n doubles through one CU, and 16 B of traffic per trailing
// entry and column -- 1230 ms at n = 8192, 63 ms at n = 2048; here 8 B, read only: the
// trailing block is updated once per panel by the MFMA GEMM, A22 -= V W^T + W V^T as
// C += [V | W] [-W | -V]^T -- 429 ms and 33 ms).
//
//   panel P1 (n x 2 nb, row-major): row r = [V(r, 0..nb) | W(r, 0..nb)]
//   k_tdb_column(j)  32 workgroups, rows r >= j:  (a) finishes column jj-1 of W from the
//       launch before it: w = w' - c v, c = tau/2 (w'^T v) summed from that launch's partials;
//       (b) a_j(r) = A(j, r) - V(r,:) W(j,:)^T - W(r,:) V(j,:)^T (A symmetric: row j is column
//       j, contiguous); (c) per-workgroup partials of ||a_j(j+2:)||^2, V^T a_j, W^T a_j.
//   k_tdb_symv(j)    one workgroup per 8 rows r >= j+1: prologue (every workgroup, identical):
//       partials -> beta, tau, scale, V^T v, W^T v (v = e_1 + scale a_j(j+2:): linear in a_j);
//       body: y(r) = A22(r,:) v, w'(r) = tau (y(r) - V(r,:) (W^T v) - W(r,:) (V^T v)), v(r) into
//       the panel and into row j of A (kept for the back-transform), partial of w'^T v.
//   k_tdb_panel_end  finishes the last W column of the panel, writes P2 = [-W | -V].
// d_j = a_j(j), e_j = beta.  Validated formula by formula against a NumPy emulation
// (tests/probes/blocked_tridiag_emulation.py).
constexpr int kTdNb = 32;
constexpr int kTdColWgs = 32;                 // workgroups of k_tdb_column (partial sets)
constexpr int kTdPartStride = 2 * kTdNb + 2;  // V^T a | W^T a | norm2 | pad
constexpr int kTdSymvRows = 2;                // rows per wave of k_tdb_symv (4 waves per workgroup)
constexpr int kTdSymvUnroll = 16 / kTdSymvRows;  // 16-byte loads per row and lane in flight

// sum of p[0 .. count) by a whole workgroup (256 or 1024 threads), identical in every thread
// and every workgroup: thread t adds p[t], p[t + T], ...; wave tree; the wave sums in order.
// scratch: >= 16 doubles of LDS; ends with a barrier (scratch may be reused afterwards).
__device__ __forceinline__ double tdb_sum_partials(const double* __restrict__ p, int count,
                                                   double* scratch) {
  double v = 0.0;
  for (int i = threadIdx.x; i < count; i += blockDim.x) v += p[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double sum = 0.0;
  for (int w = 0; w < nw; ++w) sum += scratch[w];
  __syncthreads();
  return sum;
}

// lane l of a wave holds panel entry l of a row: l < nb -> V(r, l), else W(r, l - nb)
__global__ __launch_bounds__(1024) void k_tdb_column(
    const double* __restrict__ A, int ld, int n, int j, int jj, double* __restrict__ P1,
    const double* __restrict__ wprime, const double* __restrict__ part2, int npart2,
    const double* __restrict__ taus, double* __restrict__ avec, double* __restrict__ part1) {
  __shared__ double sm[16][kTdPartStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane < kTdNb ? lane : lane - kTdNb;  // panel column of this lane
  const bool live = k < jj;                          // columns 0 .. jj-1 exist
  const bool fresh = jj > 0 && lane == kTdNb + jj - 1;  // lane of the W column being finished
  const int rows = n - j;
  const int per_wg = ((rows + gridDim.x - 1) / gridDim.x + 15) & ~15;
  const int r_begin = j + blockIdx.x * per_wg;
  const int r_end = min(n, r_begin + per_wg);
  // Everything that does not depend on c is fetched BEFORE the partial sum and its barrier
  // (the data was written by the previous launch on other XCDs: every first touch is a
  // ~1 us round trip, and the kernel is a chain of them): row j of the panel, w'(j), and the
  // first block of rows of this wave.
  const double* pj = P1 + (size_t)j * 2 * kTdNb;
  double pj_mine = 0.0, pj_vprev = 0.0, wp_j = 0.0;
  if (live) pj_mine = lane < kTdNb ? pj[kTdNb + k] : pj[k];  // W(j, k) | V(j, k)
  if (jj > 0) {
    pj_vprev = pj[jj - 1];
    wp_j = wprime[j];
  }
  constexpr int RB = 4;  // rows of a wave per iteration (independent loads in flight; 8 spill)
  double mine[RB], arow[RB], wpr[RB], vpr[RB];
  auto fetch = [&](int rb) {
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int r = rb + 16 * q;
      mine[q] = arow[q] = wpr[q] = vpr[q] = 0.0;
      if (r < r_end) {
        const double* pr = P1 + (size_t)r * 2 * kTdNb;
        if (live) mine[q] = pr[lane];
        if (fresh) {
          wpr[q] = wprime[r];
          vpr[q] = pr[jj - 1];
        }
        arow[q] = A[(size_t)j * ld + r];
      }
    }
  };
  fetch(r_begin + wave);
  // (a) c of the previous column: the sum of the symv launch's per-workgroup partials, in a
  // fixed order that is the same in every workgroup (strided per thread, wave tree, then the
  // 16 wave sums in order) -- one round of loads, not a serial chain of n / 8 of them
  double c_prev = 0.0;
  if (jj > 0) c_prev = 0.5 * taus[j - 1] * tdb_sum_partials(part2, npart2, &sm[0][0]);
  // row j of the panel, halves swapped: lane l < nb multiplies V(r, l) with W(j, l), ...
  double other = pj_mine;
  if (live && lane < kTdNb && k == jj - 1) other = wp_j - c_prev * pj_vprev;  // W(j, jj-1)
  double acc = 0.0, norm2 = 0.0;  // acc: lane l accumulates P1(r, l) * a(r), r >= j + 1
  for (int rb = r_begin + wave; rb < r_end; rb += 16 * RB) {
    if (rb != r_begin + wave) fetch(rb);
    double t[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int r = rb + 16 * q;
      if (fresh && r < r_end) {
        mine[q] = wpr[q] - c_prev * vpr[q];  // w = w' - c v
        P1[(size_t)r * 2 * kTdNb + lane] = mine[q];
      }
      t[q] = mine[q] * other;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < RB; ++q) t[q] += __shfl_xor(t[q], o);
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int r = rb + 16 * q;
      if (r >= r_end) break;  // (wave-uniform)
      const double a = arow[q] - t[q];
      if (lane == 0) avec[r] = a;
      if (r >= j + 1) acc = __builtin_fma(mine[q], a, acc);
      if (r >= j + 2) norm2 = __builtin_fma(a, a, norm2);
    }
  }
  sm[wave][lane] = acc;
  if (lane == 0) sm[wave][2 * kTdNb] = norm2;
  __syncthreads();
  if (threadIdx.x <= 2 * kTdNb) {
    double sum = 0.0;
    for (int w = 0; w < 16; ++w) sum += sm[w][threadIdx.x];
    part1[(size_t)blockIdx.x * kTdPartStride + threadIdx.x] = sum;
  }
}

// rows / columns j+1 .. n-1 of the (panel-start) matrix.  One wave = kTdSymvRows rows.
// The streaming loop does not depend on the reflector scalars: it accumulates
// z(r) = sum_{c >= j+1} A(r, c) a_j(c); y(r) = A22(r,:) v = scale z(r) + A(r, j+1) (1 - scale
// alpha) follows once the column kernel's partials have been summed.  Those partials (and
// everything else of the prologue) are REQUESTED before the loop and consumed after it, so
// their round trips hide behind the stream.
__global__ __launch_bounds__(256) void k_tdb_symv(
    double* __restrict__ A, int ld, int n, int j, int jj, double* __restrict__ P1,
    const double* __restrict__ avec, const double* __restrict__ part1,
    double* __restrict__ wprime, double* __restrict__ part2, double* __restrict__ d,
    double* __restrict__ e, double* __restrict__ taus) {
  __shared__ double sm[4][2 * kTdNb + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = j + 1;
  const int k = lane < kTdNb ? lane : lane - kTdNb;
  const bool live = k < jj;
  // ---- requests of the prologue: wave w takes the partial sets w, w + 4, ...
  constexpr int kMine = kTdColWgs / 4;
  double pl[kMine];
#pragma unroll
  for (int i = 0; i < kMine; ++i) pl[i] = part1[(size_t)(wave + 4 * i) * kTdPartStride + lane];
  const double pn = lane < kTdColWgs ? part1[(size_t)lane * kTdPartStride + 2 * kTdNb] : 0.0;
  const double alpha = avec[c0];
  const double p1 = live ? P1[(size_t)c0 * 2 * kTdNb + lane] : 0.0;
  const double dj = avec[j];
  const int r0 = c0 + (blockIdx.x * 4 + wave) * kTdSymvRows;
  int rows[kTdSymvRows];
  double acc[kTdSymvRows], arc0[kTdSymvRows], prow[kTdSymvRows], arow[kTdSymvRows];
#pragma unroll
  for (int r = 0; r < kTdSymvRows; ++r) {
    rows[r] = min(r0 + r, n - 1);
    acc[r] = 0.0;
    arc0[r] = A[(size_t)rows[r] * ld + c0];
    prow[r] = live ? P1[(size_t)rows[r] * 2 * kTdNb + lane] : 0.0;
    arow[r] = avec[rows[r]];
  }
  // ---- the stream: two rows per wave, four 16-byte loads per row and lane in flight (8 KB
  // per wave, 16 waves per CU = 128 KB per CU: what ~5 us of loaded latency needs; four rows
  // and one load each reached 2.2 TB/s, this form 5 TB/s)
  if (r0 < n) {
    const int kbeg = c0 & ~1;
    for (int cb = kbeg + 2 * lane; cb < n; cb += kTdSymvUnroll * 128) {
      double2 a[kTdSymvRows][kTdSymvUnroll];
      double x0[kTdSymvUnroll], x1[kTdSymvUnroll];
#pragma unroll
      for (int u = 0; u < kTdSymvUnroll; ++u) {
        const int c = cb + u * 128;
        const bool in = c < n;  // (c + 1 may be the padding column: masked below)
#pragma unroll
        for (int r = 0; r < kTdSymvRows; ++r)
          a[r][u] = in ? *reinterpret_cast<const double2*>(A + (size_t)rows[r] * ld + c)
                       : make_double2(0.0, 0.0);
        // (one 16-byte load of a(c), a(c + 1): c is even, avec is 16-byte aligned and padded)
        const double2 av = in ? *reinterpret_cast<const double2*>(avec + c)
                              : make_double2(0.0, 0.0);
        x0[u] = (in && c >= c0) ? av.x : 0.0;
        x1[u] = (in && c + 1 < n) ? av.y : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kTdSymvUnroll; ++u)
#pragma unroll
        for (int r = 0; r < kTdSymvRows; ++r) {
          // (x is 0 outside [c0, n): the padding past column n may hold anything, and a NaN
          //  times 0 would poison the sum, so the matrix value is masked too)
          acc[r] = __builtin_fma(x0[u] != 0.0 ? a[r][u].x : 0.0, x0[u], acc[r]);
          acc[r] = __builtin_fma(x1[u] != 0.0 ? a[r][u].y : 0.0, x1[u], acc[r]);
        }
    }
  }
  // ---- the column kernel's partials -> reflector scalars and V^T v, W^T v
  double s_w = 0.0;
#pragma unroll
  for (int i = 0; i < kMine; ++i) s_w += pl[i];
  sm[wave][lane] = s_w;
  double norm2 = pn;  // lanes 0 .. kTdColWgs-1 hold one partial each: fixed-order tree
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) norm2 += __shfl_xor(norm2, o);
  __syncthreads();
  const double s_l = ((sm[0][lane] + sm[1][lane]) + sm[2][lane]) + sm[3][lane];
  double tau = 0.0, scale = 0.0, beta = alpha;
  if (norm2 > 0.0) {
    beta = -copysign(sqrt(__builtin_fma(alpha, alpha, norm2)), alpha);
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  // t_l = (V^T v)(l) for l < nb, (W^T v)(l - nb) above: row j+1 carries v = 1, the rest scale * a
  const double t_l = live ? p1 + scale * (s_l - p1 * alpha) : 0.0;
  // lane l < nb pairs V(r, l) with (W^T v)(l): the other half's value
  const double u_l = __shfl_xor(t_l, kTdNb);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    d[j] = dj;
    e[j] = beta;
    taus[j] = tau;
  }
  double wv = 0.0;  // this wave's share of w'^T v
  if (r0 < n) {
    const double unit = 1.0 - scale * alpha;  // v(j+1) = 1 instead of scale * a(j+1)
#pragma unroll
    for (int r = 0; r < kTdSymvRows; ++r) {
      if (r0 + r >= n) break;  // (wave-uniform)
      const int row = r0 + r;
      double z = acc[r];
      double corr = prow[r] * u_l;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        z += __shfl_xor(z, o);
        corr += __shfl_xor(corr, o);
      }
      const double y = __builtin_fma(scale, z, arc0[r] * unit);
      const double vr = row == c0 ? 1.0 : scale * arow[r];
      const double wp = tau * (y - corr);
      if (lane == 0) {
        wprime[row] = wp;
        P1[(size_t)row * 2 * kTdNb + jj] = vr;        // column jj of V
        A[(size_t)j * ld + row] = vr;                  // reflector j, kept for the back-transform
        wv = __builtin_fma(wp, vr, wv);
      }
    }
  }
  __syncthreads();  // (sm is reused)
  if (lane == 0) sm[wave][0] = wv;
  __syncthreads();
  if (threadIdx.x == 0) part2[blockIdx.x] = ((sm[0][0] + sm[1][0]) + sm[2][0]) + sm[3][0];
}

__global__ void k_tdb_last(const double* __restrict__ avec, int n, double* __restrict__ d,
                           double* __restrict__ e, double* __restrict__ taus) {
  d[n - 1] = avec[n - 1];
  e[n - 1] = 0.0;
  taus[n - 1] = 0.0;
}

// end of a panel of `nbk` columns (j1 = first row / column behind it): the last W column,
// then P2(r, :) = [-W(r, :) | -V(r, :)] for rows r >= j1 (columns >= nbk of a short last
// panel are zeroed in both so that the K = 2 nb product ignores them)
__global__ __launch_bounds__(256) void k_tdb_panel_end(
    int n, int j1, int nbk, double* __restrict__ P1, double* __restrict__ P2,
    const double* __restrict__ wprime, const double* __restrict__ part2, int npart2,
    const double* __restrict__ taus) {
  __shared__ double scratch[16];
  const double c = 0.5 * taus[j1 - 1] * tdb_sum_partials(part2, npart2, scratch);
  const int lane = threadIdx.x & 63;
  const int r = j1 + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  double* pr = P1 + (size_t)r * 2 * kTdNb;
  const int k = lane < kTdNb ? lane : lane - kTdNb;
  double val = 0.0;
  if (k < nbk) {
    val = pr[lane];
    if (lane == kTdNb + nbk - 1) val = wprime[r] - c * pr[nbk - 1];
  }
  pr[lane] = val;
  P2[(size_t)r * 2 * kTdNb + (lane ^ kTdNb)] = -val;
}

// ---------------------------------------------------------------- bisection
// bounds[0..2] = {gl, gu, pivmin}; e2[i] = e[i]^2
__global__ __launch_bounds__(1024) void k_td_bounds(const double* __restrict__ d,
                                                    const double* __restrict__ e, int n,
                                                    double* __restrict__ e2,
                                                    double* __restrict__ bounds) {
  __shared__ double smin[16], smax[16], se[16];
  const int tid = threadIdx.x;
  double lo = __builtin_huge_val(), hi = -__builtin_huge_val(), emax = 0.0;
  for (int i = tid; i < n; i += 1024) {
    const double el = i > 0 ? fabs(e[i - 1]) : 0.0;
    const double er = i < n - 1 ? fabs(e[i]) : 0.0;
    lo = fmin(lo, d[i] - el - er);
    hi = fmax(hi, d[i] + el + er);
    if (i < n - 1) {
      const double sq = e[i] * e[i];
      e2[i] = sq;
      emax = fmax(emax, sq);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o));
    hi = fmax(hi, __shfl_xor(hi, o));
    emax = fmax(emax, __shfl_xor(emax, o));
  }
  if ((tid & 63) == 0) {
    smin[tid >> 6] = lo;
    smax[tid >> 6] = hi;
    se[tid >> 6] = emax;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w2 = 1; w2 < 16; ++w2) {
      lo = fmin(lo, smin[w2]);
      hi = fmax(hi, smax[w2]);
      emax = fmax(emax, se[w2]);
    }
    const double tnorm = fmax(fabs(lo), fabs(hi));
    // far above the underflow threshold so that e2 / q stays finite (a reciprocal with
    // Newton steps is used, not an IEEE division); 1e-280 * scale is still ~1e-264 relative
    const double pivmin = 1e-280 * fmax(1.0, emax);
    const double slack = 2.1 * tnorm * 2.220446049250313e-16 * n + 2.1 * pivmin;
    bounds[0] = lo - slack;
    bounds[1] = hi + slack;
    bounds[2] = pivmin;
  }
}

__device__ __forceinline__ double td_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return r;
}

// number of eigenvalues of T that are <= x (dstebz's Sturm count).  d / e2 are read in blocks
// of 8 ahead of the 8 dependent steps that use them (wave-uniform addresses: scalar loads the
// compiler merges) -- one load per step inside the chain cost ~300 cycles per step.
__device__ __forceinline__ int td_count(const double* __restrict__ d,
                                        const double* __restrict__ e2, int n, double x,
                                        double pivmin) {
  double q = d[0] - x;
  if (fabs(q) < pivmin) q = -pivmin;
  int count = q <= 0.0;
  int i = 1;
  for (; i + 8 <= n; i += 8) {
    double dd[8], ee[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      dd[u] = d[i + u];
      ee[u] = e2[i + u - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      q = (dd[u] - x) - ee[u] * td_rcp(q);
      if (fabs(q) < pivmin) q = -pivmin;
      count += q <= 0.0;
    }
  }
  for (; i < n; ++i) {
    q = (d[i] - x) - e2[i - 1] * td_rcp(q);
    if (fabs(q) < pivmin) q = -pivmin;
    count += q <= 0.0;
  }
  return count;
}

// Four lanes per eigenvalue: each round evaluates the Sturm count at the four interior points
// (quarters and 3/8), which shrinks it at least 4x (2 bits) for the latency of ONE
// count -- a count is a serial chain of n divisions, and with one thread per eigenvalue only
// n / 64 waves exist to hide it (35 ms at n = 8192 as plain bisection, 56 rounds; 28 rounds
// here on 4x the waves).  k-th smallest eigenvalue, written DESCENDING (theta[n - 1 - k]).
__global__ __launch_bounds__(64) void k_td_bisect(const double* __restrict__ d,
                                                  const double* __restrict__ e2, int n,
                                                  const double* __restrict__ bounds,
                                                  double* __restrict__ theta_desc) {
  const int lane = threadIdx.x;
  const int sub = lane & 3;
  const int k = blockIdx.x * 16 + (lane >> 2);
  const int kk = min(k, n - 1);  // keep the wave converged: loop-uniform loads of d / e2
  double lo = bounds[0], hi = bounds[1];
  const double pivmin = bounds[2];
  for (int it = 0; it < 48; ++it) {
    const double w = hi - lo;
    const double mid = 0.5 * (lo + hi);
    // the cut points: quarters (the midpoint exactly as bisection computes it, so that the
    // bracket keeps shrinking down to neighbouring doubles) and one more at 3/8
    const double x = sub == 1 ? mid : lo + w * (sub == 0 ? 0.25 : (sub == 2 ? 0.75 : 0.375));
    const bool stuck = !(mid > lo && mid < hi);
    if (__all(stuck)) break;
    const int cnt = td_count(d, e2, n, x, pivmin);
    const int base = lane & ~3;
    double xs[4];
    int cs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      xs[q] = __shfl(x, base + q);
      cs[q] = __shfl(cnt, base + q);
    }
    if (!stuck) {
      // eigenvalue kk lies right of every cut with fewer than kk + 1 eigenvalues at or below it
      double nlo = lo, nhi = hi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (cs[q] < kk + 1) {
          if (xs[q] > nlo) nlo = xs[q];
        } else if (xs[q] < nhi) {
          nhi = xs[q];
        }
      }
      lo = nlo;
      hi = nhi;
    }
  }
  if (k < n && sub == 0) theta_desc[n - 1 - k] = 0.5 * (lo + hi);
}

// ---------------------------------------------------------------- back-transform
// Z[:, q] <- Q Z[:, q],  Q = H_0 H_1 ... H_{n-2},  H_j = I - tau_j v_j v_j^T (LAPACK dormtr,
// unblocked): one workgroup per column, the column in LDS (n <= kTdLdsRows) or in place in
// global memory, reflectors applied last to first; each is one contiguous row of A.  The
// workgroups of all columns walk the rows in step, so a row is fetched from HBM once.
constexpr int kTdLdsRows = 16384;

template <bool LDS>
__global__ __launch_bounds__(1024) void k_td_backtransform(
    const double* __restrict__ A, int ld, int n, const double* __restrict__ taus,
    double* __restrict__ Z, int ldz) {
  extern __shared__ double td_x[];
  __shared__ double sm[16];
  const int tid = threadIdx.x;
  double* zg = Z + (size_t)blockIdx.x * ldz;
  double* x = LDS ? td_x : zg;
  if (LDS) {
    for (int i = tid; i < n; i += 1024) x[i] = zg[i];
  }
  __syncthreads();
  // thread t owns x[t], x[t + 1024], ... for every reflector: no barrier is needed for x
  // itself, only the two inside the block sum
  for (int j = n - 2; j >= 0; --j) {
    const double tau = taus[j];
    if (tau == 0.0) continue;  // (uniform)
    const double* v = A + (size_t)j * ld;
    int i0 = tid;
    if (i0 < j + 1) i0 += ((j + 1 - i0 + 1023) >> 10) << 10;
    double part = 0.0;
    for (int i = i0; i < n; i += 1024) part = __builtin_fma(v[i], x[i], part);
    const double s = tau * td_block_sum(part, sm);
    for (int i = i0; i < n; i += 1024) x[i] = __builtin_fma(-s, v[i], x[i]);
  }
  __syncthreads();
  if (LDS) {
    for (int i = tid; i < n; i += 1024) zg[i] = x[i];
  }
}

// ---------------------------------------------------------------- launchers
void launch_td_materialize(hipStream_t s, const double* S, int ld, int n, const double* c,
                           const double* p, double* M) {
  hipLaunchKernelGGL(k_td_materialize, dim3(std::min(8, (n + 255) / 256), n), dim3(256), 0,
                     s, S, ld, n, c, p, M);
}

// every eigenvalue of the tridiagonal (d, e), descending, into theta_desc[0..n).
// work: n + 4 doubles.
void launch_tridiagonal_eigenvalues(hipStream_t s, const double* d, const double* e, int n,
                                    double* theta_desc, double* work) {
  double* e2 = work;
  double* bounds = work + n;
  hipLaunchKernelGGL(k_td_bounds, dim3(1), dim3(1024), 0, s, d, e, n, e2, bounds);
  hipLaunchKernelGGL(k_td_bisect, dim3((n + 15) / 16), dim3(64), 0, s, d, e2, n, bounds,
                     theta_desc);
}

// The blocked form of the same reduction (same outputs: d, e, taus, reflector j in
// A[j, j+1 .. n-1]).  panel: 2 * n * 2 kTdNb doubles (P1 | P2); work: 2 n doubles +
// (kTdColWgs * (2 kTdNb + 2)) + n / 16 + 2 doubles.  `splitk_ws`: the GEMM's scratch.
void launch_tridiagonalize_blocked(hipStream_t s, double* A, int ld, int n, double* d, double* e,
                                   double* taus, double* panel, double* work,
                                   double* splitk_ws) {
  double* P1 = panel;
  double* P2 = panel + (size_t)n * 2 * kTdNb;
  double* avec = work;
  double* wprime = work + n;
  double* part1 = work + 2 * (size_t)n;
  double* part2 = part1 + kTdColWgs * kTdPartStride;
  int npart2 = 0;
  for (int j0 = 0; j0 < n; j0 += kTdNb) {
    const int nbk = std::min(kTdNb, n - j0);
    for (int jj = 0; jj < nbk; ++jj) {
      const int j = j0 + jj;
      const int rows = n - j;
      (void)rows;  // (always kTdColWgs workgroups: idle ones leave zero partials)
      hipLaunchKernelGGL(k_tdb_column, dim3(kTdColWgs), dim3(1024), 0, s, A, ld, n, j, jj, P1,
                         wprime, part2, npart2, taus, avec, part1);
      if (j == n - 1) {  // last diagonal entry: no reflector
        hipLaunchKernelGGL(k_tdb_last, dim3(1), dim3(1), 0, s, avec, n, d, e, taus);
        break;
      }
      npart2 = (n - j - 1 + 4 * kTdSymvRows - 1) / (4 * kTdSymvRows);
      hipLaunchKernelGGL(k_tdb_symv, dim3(npart2), dim3(256), 0, s, A, ld, n, j, jj, P1, avec,
                         part1, wprime, part2, d, e, taus);
    }
    const int j1 = j0 + nbk;
    if (j1 < n) {
      hipLaunchKernelGGL(k_tdb_panel_end, dim3((n - j1 + 3) / 4), dim3(256), 0, s, n, j1, nbk, P1,
                         P2, wprime, part2, npart2, taus);
      double* C = A + (size_t)j1 * ld + j1;
      launch_gemm_nt(s, P1 + (size_t)j1 * 2 * kTdNb, 2 * kTdNb, P2 + (size_t)j1 * 2 * kTdNb,
                     2 * kTdNb, C, ld, n - j1, n - j1, 2 * kTdNb, kEpiAdd, true, splitk_ws,
                     nullptr, nullptr, C);
    }
  }
}

// Z (column-major, `cols` columns of n, ldz) <- Q Z with the reflectors launch_tridiagonalize_blocked
// left in A / taus.
void launch_td_backtransform(hipStream_t s, const double* A, int ld, int n, const double* taus,
                             double* Z, int ldz, int cols) {
  if (cols <= 0) return;
  if (n <= kTdLdsRows) {
    const size_t lds = (size_t)n * sizeof(double);
    // above the 64 KB default of dynamic LDS: opt in once per device (handles of several
    // devices / host threads share the process)
    SC_OPT_IN_LDS(&k_td_backtransform<true>, kTdLdsRows * (int)sizeof(double));
    hipLaunchKernelGGL(k_td_backtransform<true>, dim3(cols), dim3(1024), lds, s, A, ld, n, taus,
                       Z, ldz);
  } else {
    hipLaunchKernelGGL(k_td_backtransform<false>, dim3(cols), dim3(1024), 0, s, A, ld, n, taus,
                       Z, ldz);
  }
}

}  // namespace sc
