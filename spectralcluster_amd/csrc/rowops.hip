// Row-parallel fp64 kernels of the refinement chain (reference refinement.py),
// the Laplacian (laplacian.py) and the scaling vectors of the symmetric
// eigen-operator.  One 256-thread workgroup per matrix row; 16-byte loads;
// reductions = wave64 shuffles + one LDS hop.  HBM-bound: 1 read (+ an L2 re-read
// for two-pass ops) and 1 write of the n x n matrix per op.
//
// Compiled with -ffp-contract=off: the elementwise arithmetic keeps the
// reference's operation order and rounding (no fused multiply-add).
#include <algorithm>
#include <cstring>

#include "dpp.h"
#include "sc_internal.h"

namespace sc {

constexpr int kRowThreads = 256;

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// block-wide reductions for 256 threads (4 waves); `sm` has >= 4 doubles.
__device__ __forceinline__ double block_max(double v, double* sm) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
__device__ __forceinline__ double block_sum(double v, double* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ---- A1: rows of X to unit L2 norm (utils.py:32-33) -------------------------
// One wave per row (four rows per workgroup, no LDS, no barrier: d is a few hundred).  The sum
// keeps the order of the 256-thread form it replaces -- accumulator w of lane l adds the
// elements 64 w + l + 256 i, each accumulator is reduced across the wave, then
// (s0 + s1) + (s2 + s3) -- so the norms are bit-identical to it.
__device__ __forceinline__ void normalize_rows_body(
    const double* __restrict__ X, int ldx, int n, int d, double* __restrict__ Xn,
    int* __restrict__ bad_rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* x = X + (size_t)row * ldx;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j0 = 0; j0 < d; j0 += kRowThreads) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int j = j0 + 64 * w + lane;
      if (j < d) acc[w] += x[j] * x[j];
    }
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) acc[w] = wave_sum(acc[w]);
  const double norm = sqrt((acc[0] + acc[1]) + (acc[2] + acc[3]));
  // a zero (or non-finite) row turns into NaNs below, like utils.py:33; later max-type
  // kernels would drop them where np.maximum keeps them, so the fact is recorded here
  if (bad_rows != nullptr && lane == 0 && !(norm > 0.0 && isfinite(norm))) *bad_rows = 1;
  double* o = Xn + (size_t)row * ldx;
  for (int j = lane; j < ldx; j += 64) o[j] = j < d ? x[j] / norm : 0.0;
}
__global__ __launch_bounds__(kRowThreads) void k_normalize_rows(
    const double* __restrict__ X, int ldx, int n, int d, double* __restrict__ Xn,
    int* __restrict__ bad_rows) {
  normalize_rows_body(X, ldx, n, d, Xn, bad_rows);
}
// ---- grouped forms of the row kernels (batch_group.hip): blockIdx.y = member of a batch
//      group, argument blocks by value in the kernel arguments, n = 0 = idle member
__global__ __launch_bounds__(kRowThreads) void k_normalize_rows_g(const GroupOf<FrontItem> g) {
  const FrontItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x * 4 >= a.n) return;
  normalize_rows_body(a.X, a.ldx, a.n, a.d, a.Xn, a.symflag + 1);
}
// the words the stages of a member latch into: symflag[1] (a zero / non-finite embedding
// row), flags[12] (non-finite scaling vectors), flags[13..15] (the Lanczos chain's latch)
__global__ void k_front_words_init_g(const GroupOf<FrontItem> g) {
  const FrontItem& a = g.s[blockIdx.x];
  if (a.n <= 0) return;
  if (threadIdx.x == 0 && a.symflag != nullptr) a.symflag[1] = 0;
  if (threadIdx.x >= 12 && threadIdx.x < 16) a.flags[threadIdx.x] = 0;
}

// ---- R1: CropDiagonal (refinement.py:145-151) -------------------------------
__global__ __launch_bounds__(kRowThreads) void k_crop_diagonal(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const double* x = in + (size_t)row * ld;
  double* o = out + (size_t)row * ld;
  double m = 0.0;  // the zero-filled diagonal takes part in the max
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    const double2 v = *reinterpret_cast<const double2*>(x + j);
    if (j != row) m = fmax(m, v.x);
    if (j + 1 < n && j + 1 != row) m = fmax(m, v.y);
  }
  m = block_max(m, sm);
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    double2 v = *reinterpret_cast<const double2*>(x + j);
    if (j == row) v.x = m;
    if (j + 1 == row) v.y = m;
    *reinterpret_cast<double2*>(o + j) = v;
  }
}

// ---- R1 (fused form): only the cropped diagonal value max(0, max_{j != i} a_ij) ----
__global__ __launch_bounds__(kRowThreads) void k_crop_value(
    const double* __restrict__ in, int n, int ld, double* __restrict__ dvec) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const double* x = in + (size_t)row * ld;
  double m = 0.0;
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    const double2 v = *reinterpret_cast<const double2*>(x + j);
    if (j != row) m = fmax(m, v.x);
    if (j + 1 < n && j + 1 != row) m = fmax(m, v.y);
  }
  m = block_max(m, sm);
  if (threadIdx.x == 0) dvec[row] = m;
}

// cut[i] = (max over the per-tile partial row maxima) * p   (refinement.py:188-191)
// one wave per row: lanes stride the row's partials (coalesced), wave max
__device__ __forceinline__ void cut_from_partials_body(const double* __restrict__ partials,
                                                       int n, int ntiles, double p,
                                                       double* __restrict__ cut) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  double m = -INFINITY;
  for (int t = lane; t < ntiles; t += 64) m = fmax(m, partials[(size_t)i * ntiles + t]);
  m = wave_max(m);
  if (lane == 0) cut[i] = m * p;
}
__global__ __launch_bounds__(256) void k_cut_from_partials(const double* __restrict__ partials,
                                                           int n, int ntiles, double p,
                                                           double* __restrict__ cut) {
  cut_from_partials_body(partials, n, ntiles, p, cut);
}
__global__ __launch_bounds__(256) void k_cut_from_partials_g(const GroupOf<FrontItem> g,
                                                             double p) {
  const FrontItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x * 4 >= a.n) return;
  // (p_own: the members of an AutoTune sweep differ in nothing but their p_percentile)
  cut_from_partials_body(a.rmpart, a.n, a.blur_cols, a.p_own > 0.0 ? a.p_own : p, a.cut);
}
__global__ __launch_bounds__(kRowThreads) void k_cut_from_rows(
    const double* __restrict__ in, int n, int ld, double p, double* __restrict__ cut,
    int zero_diag) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const double* x = in + (size_t)row * ld;
  double m = -INFINITY;
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    double2 v = *reinterpret_cast<const double2*>(x + j);
    if (zero_diag) {  // thresholding_preserve_diagonal: diagonal zeroed first (:185-186)
      if (j == row) v.x = 0.0;
      if (j + 1 == row) v.y = 0.0;
    }
    m = fmax(m, v.x);
    if (j + 1 < n) m = fmax(m, v.y);
  }
  m = block_max(m, sm);
  if (threadIdx.x == 0) cut[row] = m * p;
}

// ---- R3, ThresholdType.Percentile: cut[i] = np.percentile(row_i, 100 p) ------------
// (refinement.py:192-197; numpy 2.2 `_quantile`, method "linear").  One workgroup per row:
// the row is turned into order-preserving 64-bit keys in LDS and the element of rank
// `prev` is found by an 8-pass MSB radix select (256-bin LDS histograms); rank prev+1 is
// either the same value (duplicates) or the smallest larger key.  Then numpy's `_lerp`:
//   diff = b - a;  t >= 0.5 ? b - diff * (1 - t) : a + diff * t        (no fma)
__device__ __forceinline__ unsigned long long f64_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

__device__ __forceinline__ void row_percentile_cut_body(
    const double* __restrict__ in, int n, int ld, int zero_diag, int prev, int next,
    double gamma, double* __restrict__ cut, int keys_in_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long pkeys[];
  __shared__ int hist[256];
  __shared__ unsigned long long s_prefix, s_mask;
  __shared__ int s_rank;
  __shared__ unsigned long long s_min[4];
  __shared__ int s_cnt[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* x = in + (size_t)row * ld;
  auto key_at = [&](int j) -> unsigned long long {
    if (keys_in_lds) return pkeys[j];
    const double v = (zero_diag && j == row) ? 0.0 : x[j];
    return f64_key(v);
  };
  if (keys_in_lds)
    for (int j = tid; j < n; j += 256)
      pkeys[j] = f64_key((zero_diag && j == row) ? 0.0 : x[j]);
  if (tid == 0) { s_prefix = 0; s_mask = 0; s_rank = prev; }
  __syncthreads();
  for (int pass = 7; pass >= 0; --pass) {
    hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix, mask = s_mask;
    const int shift = 8 * pass;
    // Histogram with wave-aggregated atomics: affinities live in [0, 1], so in the first passes
    // (sign, exponent, leading mantissa bits) nearly every key of the row falls into one or two
    // bins and 4096-8192 LDS atomics on the same address serialise (the 16-member Turn-to-Diarize
    // sweep at n = 4096 spent 2.9 ms in this kernel, profiles/r37).  Lanes that share the wave
    // leader's bin send ONE atomicAdd of their count; after three such rounds whatever is left
    // (scattered keys: the late passes) goes one by one.
    for (int j0 = 0; j0 < n; j0 += 256) {  // (wave-uniform trip count)
      const int j = j0 + tid;
      const unsigned long long k = j < n ? key_at(j) : 0ull;
      const bool active = j < n && (k & mask) == prefix;
      const int bin = (int)((k >> shift) & 255);
      unsigned long long todo = __ballot(active);
      for (int round = 0; todo != 0ull; ++round) {
        if (round == 3) {
          if ((todo >> lane) & 1ull) atomicAdd(&hist[bin], 1);
          break;
        }
        const int leader = __ffsll((long long)todo) - 1;
        const int b0 = __shfl(bin, leader);
        const unsigned long long same = __ballot(active && bin == b0) & todo;
        if (lane == leader) atomicAdd(&hist[b0], (int)__popcll(same));
        todo &= ~same;
      }
    }
    __syncthreads();
    if (wave == 0) {  // bins 4*lane .. 4*lane+3: scan, find the bin holding the rank
      const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2],
                h3 = hist[4 * lane + 3];
      int incl = h0 + h1 + h2 + h3;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
      }
      const int excl = incl - (h0 + h1 + h2 + h3);
      const int rank = s_rank;
      if (rank >= excl && rank < incl) {
        int r = rank - excl, bin = 4 * lane;
        if (r >= h0) { r -= h0; ++bin; if (r >= h1) { r -= h1; ++bin; if (r >= h2) { r -= h2; ++bin; } } }
        s_rank = r;
        s_prefix = prefix | ((unsigned long long)bin << shift);
        s_mask = mask | (255ull << shift);
      }
    }
    __syncthreads();
  }
  const unsigned long long ka = s_prefix;  // key of rank `prev`
  // rank prev+1: same key if it has duplicates beyond `prev`, else the next larger key
  int le = 0;
  unsigned long long mn = ~0ull;
  for (int j = tid; j < n; j += 256) {
    const unsigned long long k = key_at(j);
    le += k <= ka;
    if (k > ka && k < mn) mn = k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    le += __shfl_xor(le, o);
    const unsigned long long u = __shfl_xor(mn, o);
    mn = u < mn ? u : mn;
  }
  if (lane == 0) { s_cnt[wave] = le; s_min[wave] = mn; }
  __syncthreads();
  if (tid == 0) {
    const int total_le = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    unsigned long long m2 = s_min[0];
    for (int w = 1; w < 4; ++w) m2 = s_min[w] < m2 ? s_min[w] : m2;
    const double a = key_f64(ka);
    double b = a;
    if (next != prev && total_le < prev + 2) b = key_f64(m2);
    const double diff = b - a;
    cut[row] = gamma >= 0.5 ? b - diff * (1.0 - gamma) : a + diff * gamma;
  }
}
// numpy's "linear" quantile position for q = p (np.percentile(row, 100 p)): virtual index
// (n - 1) q, its floor, the next rank and the interpolation weight (host and device: the same
// IEEE double operations, no contraction)
__host__ __device__ inline void percentile_position(int n, double p, int* prev, int* next,
                                                    double* gamma) {
  const double q = (p * 100.0) / 100.0;
  const double vi = (double)(n - 1) * q;
  *prev = (int)floor(vi);
  *next = *prev + 1;
  *gamma = vi - (double)*prev;
  if (vi >= (double)(n - 1)) { *prev = n - 1; *next = n - 1; *gamma = 0.0; }
  if (vi < 0.0) { *prev = 0; *next = 0; *gamma = 0.0; }
}
__global__ __launch_bounds__(256) void k_row_percentile_cut(
    const double* __restrict__ in, int n, int ld, int zero_diag, int prev, int next,
    double gamma, double* __restrict__ cut, int keys_in_lds) {
  row_percentile_cut_body(in, n, ld, zero_diag, prev, next, gamma, cut, keys_in_lds);
}
// the members of a group (AutoTune sweep under the Turn-to-Diarize sequence: one size, one
// p_percentile each) in one launch: B1 -> cut
__global__ __launch_bounds__(256) void k_row_percentile_cut_g(const GroupOf<FrontItem> g,
                                                              int zero_diag, int keys_in_lds) {
  const FrontItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.n) return;
  int prev, next;
  double gamma;
  percentile_position(a.n, a.p_own, &prev, &next, &gamma);
  row_percentile_cut_body(a.B1, a.n, a.ldn, zero_diag, prev, next, gamma, a.cut, keys_in_lds);
}

// generic thresholding against a precomputed per-row cut (RowMax with preserved
// diagonal, Percentile): refinement.py:185-186, 200-209
__global__ __launch_bounds__(kRowThreads) void k_row_threshold_cut(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    const double* __restrict__ cut, double mult, int binarize, int preserve_diag) {
  const int row = blockIdx.x;
  const double* x = in + (size_t)row * ld;
  double* o = out + (size_t)row * ld;
  const double c = cut[row];
  for (int j = threadIdx.x; j < n; j += kRowThreads) {
    double v = x[j];
    if (preserve_diag && j == row) v = 0.0;
    double r = v < c ? v * mult : (binarize ? 1.0 : v);
    if (preserve_diag && j == row) r = 1.0;
    o[j] = r;
  }
}

// ---- R3 + R4 fused: out = sym(thr(B), thr(B)^T) over tile PAIRS ---------------------
// thr(x; cut_i) = x < cut_i ? x * mult : (binarize ? 1 : x)   (refinement.py:200-207, no
// preserve_diagonal); sym = max or average (refinement.py:219-226).  One workgroup
// handles tiles (I, J) and (J, I), I <= J: both are read once, the symmetric result is
// computed once and written to both places (the mirror through LDS, so both stores are
// coalesced): 1 read + 1 write of n^2 for the two ops together.
constexpr int kTsTile = 64;  // tile edge of the threshold + symmetrise pass
// DIGITS: the pass also leaves what the quantiser of the matrix-free Diffuse (diffuse_free.hip,
// k_free_quantize) would compute from its result in a pass of its own: TsDigits, sc_internal.h.
__device__ __forceinline__ double ts_digits(const TsDigits& dg, double sigma, int row, int blk,
                                          int c0, double v0, double v1, bool lane0 /* the half-wave's LAST lane */) {
  int qv[2];
  double q2 = 0.0;
  const double e[2] = {v0, v1};
  signed char hb[2], lb[2];
#pragma unroll
  for (int w = 0; w < 2; ++w) {  // (the quantiser's arithmetic, word for word)
    double qd = rint(e[w] * sigma);
    if (fabs(qd) > 32639.0) dg.scal[3] = 1.0;
    qd = fmin(fmax(qd, -32639.0), 32639.0);
    const int q = (int)qd;
    const int h = (q + 128) >> 8;
    const int l = q - (h << 8);
    qv[w] = q < 0 ? -q : q;
    q2 += qd * qd;  // (integers below 2^31: every partial sum is exact)
    hb[w] = (signed char)h;
    lb[w] = (signed char)l;
  }
  signed char* dst = dg.Q + (size_t)row * dg.pitch + (size_t)blk * 128 + c0;
  *reinterpret_cast<short*>(dst) = (short)((unsigned char)hb[0] | ((unsigned short)(unsigned char)hb[1] << 8));
  *reinterpret_cast<short*>(dst + 64) = (short)((unsigned char)lb[0] | ((unsigned short)(unsigned char)lb[1] << 8));
  // the 32 lanes of a half-wave hold the row's 64 columns: their sums land in its last lane
  const double ys = half_sum_to_last(v0 + v1);
  const int rs = half_sum_to_last(qv[0] + qv[1]);
  const double q2s = half_sum_to_last(q2);
  if (lane0) {
    dg.ypart[(size_t)blk * (kTsTile * dg.nblk) + row] = ys;
    dg.rpart[(size_t)blk * (kTsTile * dg.nblk) + row] = rs;
    dg.q2part[(size_t)blk * (kTsTile * dg.nblk) + row] = q2s;
  }
  return q2s;  // (valid in the half-wave's last lane)
}
__device__ __forceinline__ void ts_store_mx(const TsDigits& dg, const double* sm, int group,
                                            int blk) {
  double m = sm[0];
#pragma unroll
  for (int u = 1; u < 8; ++u) m = fmax(m, sm[u]);
  // sqrt rounds to nearest: one ulp up makes it an upper bound of the segment norm
  dg.mx64[(size_t)group * dg.nblk + blk] = sqrt(m) * (1.0 + 0x1p-52);
}
template <bool DIGITS>
__device__ __forceinline__ void threshold_symmetrize_body(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    const double* __restrict__ cut, double mult, int binarize, int symtype, int ntiles,
    int preserve_diag, const TsDigits dg) {
  // One workgroup = the tile pair (I, J), (J, I) of 64 x 64 entries.  O = sym(thr(A), thr(B)^T)
  // is tile (I, J) of the result and O^T is tile (J, I) (max and average are symmetric in their
  // arguments), so B and then O take one trip each through the one LDS tile.  Every thread
  // owns 8 x 2 entries of a tile and requests all 16 of its 16-byte loads before it touches the
  // first (round 3's 32 x 32 form had two 8-byte loads in flight per thread and 256-byte row
  // segments: 4.7 TB/s).
  __shared__ double tT[kTsTile][kTsTile + 1];
  __shared__ double smax[2][8];
  double q2m = 0.0;
  // (grouped launches carry members with and without digits: a workgroup-uniform branch)
  const bool digits = DIGITS && dg.Q != nullptr;
  // tile pair of this workgroup: row ti of the upper triangle starts at
  // off(ti) = ti * ntiles - ti (ti - 1) / 2.  Closed form + one correction step (a counting
  // loop here was 5.6e7 scalar instructions per launch at n = 8192: up to 256 trips per wave)
  const int id = blockIdx.x;
  const double b2 = 2.0 * ntiles + 1.0;
  int ti = (int)((b2 - sqrt(b2 * b2 - 8.0 * id)) * 0.5);
  ti = ti < 0 ? 0 : (ti >= ntiles ? ntiles - 1 : ti);
  auto off = [&](int t) { return t * ntiles - (t * (t - 1)) / 2; };
  while (ti > 0 && off(ti) > id) --ti;
  while (ti + 1 < ntiles && off(ti + 1) <= id) ++ti;
  const int tj = ti + (id - off(ti));
  const int bi = ti * kTsTile, bj = tj * kTsTile;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 column pairs x 8 rows
  const int c0 = 2 * tx;
  const bool diag_tile = ti == tj;
  // (rows are padded to ld, a multiple of 16 doubles: a pair that starts inside a row's
  //  storage stays inside it)
  double2 a[8], b[8];
  double ca[8], cb[8];  // the rows' cuts (requested with the tiles, not one wait each later)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = ty + 8 * q;
    const int gi = bi + r, gj = bj + c0;
    a[q] = (gi < n && gj < n) ? *reinterpret_cast<const double2*>(in + (size_t)gi * ld + gj)
                              : make_double2(0.0, 0.0);
    ca[q] = gi < n ? cut[gi] : 0.0;
  }
  if (!diag_tile) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = ty + 8 * q;
      const int gi = bj + r, gj = bi + c0;
      b[q] = (gi < n && gj < n) ? *reinterpret_cast<const double2*>(in + (size_t)gi * ld + gj)
                                : make_double2(0.0, 0.0);
      cb[q] = gi < n ? cut[gi] : 0.0;
    }
  }
  auto thr = [&](double v, double c, int gi, int gj) {
    if (!(gi < n && gj < n)) return 0.0;
    v = v < c ? v * mult : (binarize ? 1.0 : v);
    if (preserve_diag && gi == gj) v = 1.0;  // diagonal restored to 1 (:208-209)
    return v;
  };
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = ty + 8 * q;
    a[q].x = thr(a[q].x, ca[q], bi + r, bj + c0);
    a[q].y = thr(a[q].y, ca[q], bi + r, bj + c0 + 1);
    if (diag_tile) {
      b[q] = a[q];
    } else {
      b[q].x = thr(b[q].x, cb[q], bj + r, bi + c0);
      b[q].y = thr(b[q].y, cb[q], bj + r, bi + c0 + 1);
    }
    tT[r][c0] = b[q].x;
    tT[r][c0 + 1] = b[q].y;
  }
  __syncthreads();
  double sigma = 0.0;
  if (digits) {
    const double amax = dg.scal[0];
    sigma = (amax > 0.0 && isfinite(amax)) ? 32639.0 / amax : 0.0;
  }
  // O (r, c) = sym(thr A (r, c), thr B (c, r)), kept in a[]
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = ty + 8 * q;
    const double b0 = tT[c0][r], b1 = tT[c0 + 1][r];
    a[q].x = symtype == SC_SYMMETRIZE_MAX ? fmax(a[q].x, b0) : 0.5 * (a[q].x + b0);
    a[q].y = symtype == SC_SYMMETRIZE_MAX ? fmax(a[q].y, b1) : 0.5 * (a[q].y + b1);
    const int gi = bi + r, gj = bj + c0;
    if (gi < n && gj + 1 < n)
      *reinterpret_cast<double2*>(out + (size_t)gi * ld + gj) = a[q];
    else if (gi < n && gj < n)
      out[(size_t)gi * ld + gj] = a[q].x;
    // (entries outside the matrix are zero here: thr() returned 0 for them and sym(0, 0) = 0)
    if (digits) q2m = fmax(q2m, ts_digits(dg, sigma, gi, tj, c0, a[q].x, a[q].y, tx == 31));
  }
  // the tile's largest squared segment norm: 8 half-waves x 8 rows each
  if (digits && tx == 31) smax[0][ty] = q2m;
  if (diag_tile) {
    if (digits) {
      __syncthreads();
      if (threadIdx.x == 0) ts_store_mx(dg, smax[0], ti, tj);
    }
    return;
  }
  __syncthreads();  // everybody has read B^T
  if (digits && threadIdx.x == 0) ts_store_mx(dg, smax[0], ti, tj);
  q2m = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = ty + 8 * q;
    tT[r][c0] = a[q].x;
    tT[r][c0 + 1] = a[q].y;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) {  // tile (J, I) = O^T
    const int r = ty + 8 * q;
    const int gi = bj + r, gj = bi + c0;
    const double2 o = make_double2(tT[c0][r], tT[c0 + 1][r]);
    if (gi < n && gj + 1 < n)
      *reinterpret_cast<double2*>(out + (size_t)gi * ld + gj) = o;
    else if (gi < n && gj < n)
      out[(size_t)gi * ld + gj] = o.x;
    if (digits) q2m = fmax(q2m, ts_digits(dg, sigma, gi, ti, c0, o.x, o.y, tx == 31));
  }
  if (digits) {
    if (tx == 31) smax[1][ty] = q2m;
    __syncthreads();
    if (threadIdx.x == 0) ts_store_mx(dg, smax[1], tj, ti);
  }
}
__global__ __launch_bounds__(256) void k_threshold_symmetrize(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    const double* __restrict__ cut, double mult, int binarize, int symtype, int ntiles,
    int preserve_diag) {
  threshold_symmetrize_body<false>(in, out, n, ld, cut, mult, binarize, symtype, ntiles,
                                   preserve_diag, TsDigits{});
}
__global__ __launch_bounds__(256) void k_threshold_symmetrize_digits(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    const double* __restrict__ cut, double mult, int binarize, int symtype, int ntiles,
    int preserve_diag, const TsDigits dg) {
  threshold_symmetrize_body<true>(in, out, n, ld, cut, mult, binarize, symtype, ntiles,
                                  preserve_diag, dg);
}
// y1 = rowsum(A), R = sum |q| and its maximum from the per-block partials ([block][row]).  A
// workgroup takes 64 rows; thread (row, quarter) adds a quarter of the row's blocks in order,
// the quarters are folded in order, and the workgroup sends ONE candidate for the maximum (a
// wave per row and a look at the word each cost 40 us of same-address traffic for 8 us of work).
// With `segs` (the tile skip list of the digit product, sc_internal.h FreeSegs): the group's
// smallest diagonal-only candidate threshold from the rows' exact T_ii (the segment maxima mx64
// were stored by the threshold pass itself).
__device__ __forceinline__ void free_partials_reduce_body(
    const double* __restrict__ ypart, const int* __restrict__ rpart, int n, int nblk,
    double* __restrict__ y1, double* __restrict__ R, unsigned long long* __restrict__ rmax_bits,
    const FreeSegs segs) {
  __shared__ double sy[4][64];
  __shared__ double sq[4][64];
  __shared__ long long sr[4][64];
  const int r = threadIdx.x & 63, qd = threadIdx.x >> 6;
  const int row = blockIdx.x * 64 + r;
  const size_t rows = (size_t)kTsTile * nblk;
  const int b0 = (int)((long long)nblk * qd / 4), b1 = (int)((long long)nblk * (qd + 1) / 4);
  double ys = 0.0, q2s = 0.0;
  long long rs = 0;
  if (segs.q2part == nullptr) {
#pragma unroll 8
    for (int b = b0; b < b1; ++b) {
      ys += ypart[(size_t)b * rows + row];
      rs += rpart[(size_t)b * rows + row];
    }
  } else {
#pragma unroll 8
    for (int b = b0; b < b1; ++b) {
      ys += ypart[(size_t)b * rows + row];
      rs += rpart[(size_t)b * rows + row];
      q2s += row < n ? segs.q2part[(size_t)b * rows + row] : 0.0;  // (integers: exact)
    }
  }
  sy[qd][r] = ys;
  sq[qd][r] = q2s;
  sr[qd][r] = rs;
  __syncthreads();
  if (qd == 0) {
    ys = ((sy[0][r] + sy[1][r]) + sy[2][r]) + sy[3][r];
    rs = sr[0][r] + sr[1][r] + sr[2][r] + sr[3][r];
    const double rd = (double)rs;
    if (row < n) {
      y1[row] = ys;
      R[row] = rd;
    }
    double m = row < n ? rd : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if (r == 0) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
      const unsigned long long cur =
          __hip_atomic_load(rmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (bits > cur) atomicMax(rmax_bits, bits);
    }
    if (segs.q2part != nullptr) {
      // T_ii = ||q_i||^2 exactly (< 2^46); (float) of it is what the product's epilogue stores
      // for the diagonal entry, and the row's maximum M_i is at least that
      const double tii = ((sq[0][r] + sq[1][r]) + sq[2][r]) + sq[3][r];
      float t = row < n ? free_threshold((float)tii, rd, 32639.0 * (double)n, n) : INFINITY;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t = fminf(t, __shfl_xor(t, o));
      if (r == 0) segs.tau64[blockIdx.x] = t;
    }
  }
}
__global__ __launch_bounds__(256) void k_free_partials_reduce(
    const double* __restrict__ ypart, const int* __restrict__ rpart, int n, int nblk,
    double* __restrict__ y1, double* __restrict__ R, unsigned long long* __restrict__ rmax_bits,
    const FreeSegs segs) {
  free_partials_reduce_body(ypart, rpart, n, nblk, y1, R, rmax_bits, segs);
}
__global__ __launch_bounds__(256) void k_free_partials_reduce_g(const GroupOf<FreeItem> g) {
  const FreeItem& a = g.s[blockIdx.y];
  const int nblk = (a.n + kTsTile - 1) / kTsTile;
  if (a.n <= 0 || a.ypart == nullptr || (int)blockIdx.x >= nblk) return;
  free_partials_reduce_body(a.ypart, a.rpart, a.n, nblk, a.y1, a.R,
                            reinterpret_cast<unsigned long long*>(a.scal) + 2,
                            FreeSegs{a.q2part, nullptr, a.tau64});
}
__global__ __launch_bounds__(256) void k_threshold_symmetrize_g(const GroupOf<FrontItem> g,
                                                                double mult, int binarize,
                                                                int symtype, int preserve_diag) {
  const FrontItem& a = g.s[blockIdx.y];
  const int t = (a.n + kTsTile - 1) / kTsTile;
  if ((int)blockIdx.x >= t * (t + 1) / 2) return;
  threshold_symmetrize_body<false>(a.B1, a.B2, a.n, a.ldn, a.cut, mult, binarize, symtype, t,
                                   preserve_diag, TsDigits{});
}
// ... with the digits of the members that take the matrix-free Diffuse (dg.s[member].Q != nullptr)
__global__ __launch_bounds__(256) void k_threshold_symmetrize_digits_g(
    const GroupOf<FrontItem> g, const GroupOf<TsDigits> dg, double mult, int binarize, int symtype,
    int preserve_diag) {
  const FrontItem& a = g.s[blockIdx.y];
  const int t = (a.n + kTsTile - 1) / kTsTile;
  if ((int)blockIdx.x >= t * (t + 1) / 2) return;
  threshold_symmetrize_body<true>(a.B1, a.B2, a.n, a.ldn, a.cut, mult, binarize, symtype, t,
                                  preserve_diag, dg.s[blockIdx.y]);
}

// ---- R3: RowWiseThreshold, RowMax (refinement.py:182-210) --------------------
__global__ __launch_bounds__(kRowThreads) void k_row_threshold(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    double p, double mult, int binarize, int preserve_diag) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const double* x = in + (size_t)row * ld;
  double* o = out + (size_t)row * ld;
  double m = -INFINITY;
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    double2 v = *reinterpret_cast<const double2*>(x + j);
    if (preserve_diag) {  // diagonal zero-filled before the max (:185-186)
      if (j == row) v.x = 0.0;
      if (j + 1 == row) v.y = 0.0;
    }
    m = fmax(m, v.x);
    if (j + 1 < n) m = fmax(m, v.y);
  }
  m = block_max(m, sm);
  const double cut = m * p;  // row_max * p_percentile (:191)
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    double2 v = *reinterpret_cast<const double2*>(x + j);
    if (preserve_diag) {
      if (j == row) v.x = 0.0;
      if (j + 1 == row) v.y = 0.0;
    }
    // x*(!small) + x*mult*small == select (exact for finite x)
    double2 r;
    r.x = v.x < cut ? v.x * mult : (binarize ? 1.0 : v.x);
    r.y = v.y < cut ? v.y * mult : (binarize ? 1.0 : v.y);
    if (preserve_diag) {  // diagonal back to 1 (:208-209)
      if (j == row) r.x = 1.0;
      if (j + 1 == row) r.y = 1.0;
    }
    *reinterpret_cast<double2*>(o + j) = r;
  }
}

// ---- R6: RowWiseNormalize (refinement.py:240-245) ------------------------------
__global__ __launch_bounds__(kRowThreads) void k_row_normalize(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const double* x = in + (size_t)row * ld;
  double* o = out + (size_t)row * ld;
  double m = -INFINITY;
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    const double2 v = *reinterpret_cast<const double2*>(x + j);
    m = fmax(m, v.x);
    if (j + 1 < n) m = fmax(m, v.y);
  }
  m = block_max(m, sm);
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    double2 v = *reinterpret_cast<const double2*>(x + j);
    v.x = v.x / m;
    v.y = v.y / m;
    *reinterpret_cast<double2*>(o + j) = v;
  }
}

// ---- row max + row sum of the (symmetric) refined matrix ----------------------
__device__ __forceinline__ void row_stats_body(
    const double* __restrict__ in, int n, int ld, double* __restrict__ rowmax,
    double* __restrict__ rowsum) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const double* x = in + (size_t)row * ld;
  double m = -INFINITY, s = 0.0;
  for (int j = 2 * threadIdx.x; j < n; j += 2 * kRowThreads) {
    const double2 v = *reinterpret_cast<const double2*>(x + j);
    m = fmax(m, v.x);
    s += v.x;
    if (j + 1 < n) {
      m = fmax(m, v.y);
      s += v.y;
    }
  }
  m = block_max(m, sm);
  s = block_sum(s, sm);
  if (threadIdx.x == 0) {
    rowmax[row] = m;
    rowsum[row] = s;
  }
}
__global__ __launch_bounds__(kRowThreads) void k_row_stats(
    const double* __restrict__ in, int n, int ld, double* __restrict__ rowmax,
    double* __restrict__ rowsum) {
  row_stats_body(in, n, ld, rowmax, rowsum);
}
// ... of every member of a group: B2 -> rowmax, rowsum (the fields scaling_vectors_g reads)
__global__ __launch_bounds__(kRowThreads) void k_row_stats_g(const GroupOf<FrontItem> g) {
  const FrontItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.n) return;
  row_stats_body(a.B2, a.n, a.ldn, const_cast<double*>(a.rowmax), const_cast<double*>(a.rowsum));
}

// ---- scaling vectors of Op = diag(p) + diag(c) S diag(c)  ---------------------
// W = diag(a) S with a = 1/rowmax after RowWiseNormalize (a = 1 otherwise).
//   None/Affinity : Op = D_c S D_c            c = sqrt(a)            t = sqrt(a)
//   Unnormalized  : Op = -(D_deg - D_c S D_c) c = sqrt(a)            t = sqrt(a)
//   RandomWalk    : g = 1/(deg+eps)           c = sqrt(g a), p = -g deg, t = c
//   GraphCut      : h = 1/(sqrt(deg)+eps)     c = h sqrt(a), p = -h^2 deg, t = sqrt(a)
// with deg = a * rowsum(S) (laplacian.py:41 on W).  Eigenvector of the
// reference matrix = normalise(t .* u) for an eigenvector u of Op.
__device__ __forceinline__ void scaling_vectors_body(const double* __restrict__ rowmax,
                                                     const double* __restrict__ rowsum, int n,
                                                     int lap, int rownorm,
                                                     double* __restrict__ c,
                                                     double* __restrict__ p,
                                                     double* __restrict__ t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double eps = 1e-10;  // laplacian.py:6
  const double a = rownorm ? 1.0 / rowmax[i] : 1.0;
  const double sa = sqrt(a);
  const double deg = rownorm ? rowsum[i] / rowmax[i] : rowsum[i];
  double cv = sa, pv = 0.0, tv = sa;
  if (lap == SC_LAPLACIAN_UNNORMALIZED) {
    pv = -deg;
  } else if (lap == SC_LAPLACIAN_RANDOM_WALK) {
    const double g = 1.0 / (deg + eps);
    cv = sqrt(g * a);
    pv = -(g * deg);
    tv = cv;
  } else if (lap == SC_LAPLACIAN_GRAPH_CUT) {
    const double h = 1.0 / (sqrt(deg) + eps);
    cv = h * sa;
    pv = -((h * deg) * h);
  }
  c[i] = cv;
  p[i] = pv;
  t[i] = tv;
}
// flags (may be nullptr): thread 0 also prepares the solver's flag words -- flags[12] (non-finite
// input) starts from the zero-embedding-row word of the affinity stage (symflag[1], left as it is:
// it describes the resident affinity, which later calls may solve again; nullptr: from 0) and the
// chain words flags[13..15] from 0.  Rounds 1-5 did this with a device-to-device copy and a
// fill: ~10-15 us each on the stream.
__global__ void k_scaling_vectors(const double* __restrict__ rowmax,
                                  const double* __restrict__ rowsum, int n,
                                  int lap, int rownorm, double* __restrict__ c,
                                  double* __restrict__ p, double* __restrict__ t,
                                  int* __restrict__ flags, int* __restrict__ symflag) {
  scaling_vectors_body(rowmax, rowsum, n, lap, rownorm, c, p, t);
  if (flags != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    flags[12] = symflag != nullptr ? symflag[1] : 0;
    flags[13] = 0;
    flags[14] = 0;
    flags[15] = 0;
  }
}
// scaling vectors + the finite-ness word of every member (flags[12] = a NaN embedding row or
// a non-finite scaling entry: what the single-call path gets from its copy + k_check_finite)
__global__ void k_scaling_vectors_g(const GroupOf<FrontItem> g, int lap, int rownorm) {
  const FrontItem& a = g.s[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  scaling_vectors_body(a.rowmax, a.rowsum, a.n, lap, rownorm, a.cvec, a.pvec, a.tvec);
  if (!isfinite(a.cvec[i]) || !isfinite(a.pvec[i]) ||
      (i == 0 && a.symflag != nullptr && a.symflag[1] != 0))
    a.flags[12] = 1;
}

// flag = 1 if any entry of a or b is NaN / inf (np.linalg.eig raises on such input)
__global__ void k_check_finite(const double* __restrict__ a, const double* __restrict__ b,
                               int n, int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!isfinite(a[i]) || !isfinite(b[i])) *flag = 1;
}

// ---- L1: materialised Laplacian for the stage API (laplacian.py:24-60) --------
__global__ __launch_bounds__(kRowThreads) void k_row_sum(
    const double* __restrict__ in, int n, int ld, double* __restrict__ rowsum) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const double* x = in + (size_t)row * ld;
  double s = 0.0;
  for (int j = threadIdx.x; j < n; j += kRowThreads) s += x[j];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) rowsum[row] = s;
}

__global__ __launch_bounds__(kRowThreads) void k_laplacian(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    int lap, const double* __restrict__ deg) {
  const int row = blockIdx.x;
  const double eps = 1e-10;
  const double* x = in + (size_t)row * ld;
  double* o = out + (size_t)row * ld;
  const double di = deg[row];
  double ri = 1.0;
  if (lap == SC_LAPLACIAN_RANDOM_WALK) ri = 1.0 / (di + eps);
  if (lap == SC_LAPLACIAN_GRAPH_CUT) ri = 1.0 / (sqrt(di) + eps);
  for (int j = threadIdx.x; j < n; j += kRowThreads) {
    double l = (j == row ? di : 0.0) - x[j];  // degree - affinity (:42)
    if (lap == SC_LAPLACIAN_RANDOM_WALK) {
      l = ri * l;
    } else if (lap == SC_LAPLACIAN_GRAPH_CUT) {
      const double rj = 1.0 / (sqrt(deg[j]) + eps);
      l = (ri * l) * rj;  // degree_norm.dot(laplacian).dot(degree_norm) (:57)
    }
    o[j] = l;
  }
}

// ---- R4: Symmetrize (refinement.py:219-226): 32x32 tiles through LDS ----------
__global__ __launch_bounds__(256) void k_symmetrize(const double* __restrict__ in,
                                                    double* __restrict__ out,
                                                    int n, int ld, int type) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.y * 32, bj = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  // transposed tile: rows bj.., cols bi..
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int gi = bj + r, gj = bi + tx;
    tile[r][tx] = (gi < n && gj < n) ? in[(size_t)gi * ld + gj] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int gi = bi + r, gj = bj + tx;
    if (gi < n && gj < n) {
      const double a = in[(size_t)gi * ld + gj];
      const double b = tile[tx][r];
      out[(size_t)gi * ld + gj] =
          type == SC_SYMMETRIZE_MAX ? fmax(a, b) : 0.5 * (a + b);
    }
  }
}

// -------------------------------------------------------------------------------
void launch_normalize_rows(hipStream_t s, const double* X, int ldx, int n, int d,
                           double* Xn, int* bad_rows) {
  hipLaunchKernelGGL(k_normalize_rows, dim3((n + 3) / 4), dim3(kRowThreads), 0, s, X, ldx, n,
                     d, Xn, bad_rows);
}
void launch_crop_diagonal(hipStream_t s, const double* in, double* out, int n,
                          int ld) {
  hipLaunchKernelGGL(k_crop_diagonal, dim3(n), dim3(kRowThreads), 0, s, in, out, n,
                     ld);
}
void launch_row_threshold(hipStream_t s, const double* in, double* out, int n,
                          int ld, double p, double mult, int binarize,
                          int preserve_diag) {
  hipLaunchKernelGGL(k_row_threshold, dim3(n), dim3(kRowThreads), 0, s, in, out, n,
                     ld, p, mult, binarize, preserve_diag);
}
void launch_crop_value(hipStream_t s, const double* in, int n, int ld, double* dvec) {
  hipLaunchKernelGGL(k_crop_value, dim3(n), dim3(kRowThreads), 0, s, in, n, ld, dvec);
}
void launch_cut_from_partials(hipStream_t s, const double* partials, int n, int ntiles,
                              double p, double* cut) {
  hipLaunchKernelGGL(k_cut_from_partials, dim3((n + 3) / 4), dim3(256), 0, s, partials, n,
                     ntiles, p, cut);
}
void launch_cut_from_rows(hipStream_t s, const double* in, int n, int ld, double p,
                          double* cut, int zero_diag) {
  hipLaunchKernelGGL(k_cut_from_rows, dim3(n), dim3(kRowThreads), 0, s, in, n, ld, p, cut,
                     zero_diag);
}
// cut[i] = np.percentile(row_i (diagonal zeroed if asked), 100 * p)
void launch_cut_percentile(hipStream_t s, const double* in, int n, int ld, double p,
                           double* cut, int zero_diag) {
  int prev, next;
  double gamma;
  percentile_position(n, p, &prev, &next, &gamma);
  const size_t bytes = (size_t)n * sizeof(unsigned long long);
  const int in_lds = bytes <= 128 * 1024;
  SC_OPT_IN_LDS(k_row_percentile_cut, 128 * 1024);
  hipLaunchKernelGGL(k_row_percentile_cut, dim3(n), dim3(256), in_lds ? bytes : 0, s, in, n,
                     ld, zero_diag, prev, next, gamma, cut, in_lds);
}
void launch_row_threshold_cut(hipStream_t s, const double* in, double* out, int n, int ld,
                              const double* cut, double mult, int binarize,
                              int preserve_diag) {
  hipLaunchKernelGGL(k_row_threshold_cut, dim3(n), dim3(kRowThreads), 0, s, in, out, n, ld,
                     cut, mult, binarize, preserve_diag);
}
void launch_threshold_symmetrize(hipStream_t s, const double* in, double* out, int n, int ld,
                                 const double* cut, double mult, int binarize, int symtype,
                                 int preserve_diag) {
  const int t = (n + kTsTile - 1) / kTsTile;
  hipLaunchKernelGGL(k_threshold_symmetrize, dim3(t * (t + 1) / 2), dim3(256), 0, s, in, out,
                     n, ld, cut, mult, binarize, symtype, t, preserve_diag);
}
// the same pass + digits and row partials for the matrix-free Diffuse (TsDigits above); Q has
// round_up(n, 128) rows of `pitch` = 2 round_up(n, 64) bytes, the partials round_up(n, 64) rows
void launch_threshold_symmetrize_digits(hipStream_t s, const double* in, double* out, int n, int ld,
                                        const double* cut, double mult, int binarize, int symtype,
                                        int preserve_diag, signed char* Q, double* scal,
                                        double* ypart, int* rpart, double* q2part,
                                        double* mx64) {
  const int t = (n + kTsTile - 1) / kTsTile;
  const size_t pitch = (size_t)2 * t * kTsTile;
  // rows [64 t, round_up(n, 128)) belong to no tile: zero digits
  const int rows_padded = (n + 127) / 128 * 128;
  if (rows_padded > t * kTsTile)
    hipMemsetAsync(Q + (size_t)t * kTsTile * pitch, 0, (size_t)(rows_padded - t * kTsTile) * pitch, s);
  const TsDigits dg{Q, pitch, t, scal, ypart, rpart, q2part, mx64};
  hipLaunchKernelGGL(k_threshold_symmetrize_digits, dim3(t * (t + 1) / 2), dim3(256), 0, s, in,
                     out, n, ld, cut, mult, binarize, symtype, t, preserve_diag, dg);
}
void launch_free_partials_reduce(hipStream_t s, const double* ypart, const int* rpart, int n,
                                 double* y1, double* R, double* scal, const FreeSegs& segs) {
  const int nblk = (n + kTsTile - 1) / kTsTile;
  hipLaunchKernelGGL(k_free_partials_reduce, dim3(nblk), dim3(256), 0, s, ypart, rpart, n,
                     nblk, y1, R, reinterpret_cast<unsigned long long*>(scal) + 2, segs);
}
void launch_row_normalize(hipStream_t s, const double* in, double* out, int n,
                          int ld) {
  hipLaunchKernelGGL(k_row_normalize, dim3(n), dim3(kRowThreads), 0, s, in, out, n,
                     ld);
}
void launch_row_stats(hipStream_t s, const double* in, int n, int ld,
                      double* rowmax, double* rowsum) {
  hipLaunchKernelGGL(k_row_stats, dim3(n), dim3(kRowThreads), 0, s, in, n, ld,
                     rowmax, rowsum);
}
void launch_scaling_vectors(hipStream_t s, const double* rowmax,
                            const double* rowsum, int n, int laplacian_type,
                            int row_normalized, double* c, double* p, double* t, int* flags,
                            int* symflag) {
  hipLaunchKernelGGL(k_scaling_vectors, dim3((n + 255) / 256), dim3(256), 0, s,
                     rowmax, rowsum, n, laplacian_type, row_normalized, c, p, t, flags, symflag);
}
// ---- grouped launches of the stages between the two GEMMs of a batch group
static GroupOf<FrontItem> front_pack(const FrontItem* items, int count, int* nmax) {
  GroupOf<FrontItem> g;
  memset(&g, 0, sizeof(g));
  *nmax = 0;
  for (int z = 0; z < count; ++z) {
    g.s[z] = items[z];
    *nmax = std::max(*nmax, items[z].n);
  }
  return g;
}
void launch_front_begin_group(hipStream_t s, const FrontItem* items, int count,
                              bool normalize_rows) {
  int nmax;
  const GroupOf<FrontItem> g = front_pack(items, count, &nmax);
  if (nmax == 0) return;
  hipLaunchKernelGGL(k_front_words_init_g, dim3(count), dim3(64), 0, s, g);
  if (normalize_rows)
    hipLaunchKernelGGL(k_normalize_rows_g, dim3((nmax + 3) / 4, count), dim3(kRowThreads), 0, s,
                       g);
}
// Percentile cuts (p_own per member) and, after the threshold pass, row maxima / sums of B2 for
// every member of a group in one launch each
void launch_cut_percentile_group(hipStream_t s, const FrontItem* items, int count, int zero_diag) {
  int nmax;
  const GroupOf<FrontItem> g = front_pack(items, count, &nmax);
  if (nmax == 0) return;
  const size_t bytes = (size_t)nmax * sizeof(unsigned long long);
  const int in_lds = bytes <= 128 * 1024;
  SC_OPT_IN_LDS(k_row_percentile_cut_g, 128 * 1024);
  hipLaunchKernelGGL(k_row_percentile_cut_g, dim3(nmax, count), dim3(256), in_lds ? bytes : 0, s,
                     g, zero_diag, in_lds);
}
void launch_row_stats_group(hipStream_t s, const FrontItem* items, int count) {
  int nmax;
  const GroupOf<FrontItem> g = front_pack(items, count, &nmax);
  if (nmax == 0) return;
  hipLaunchKernelGGL(k_row_stats_g, dim3(nmax, count), dim3(kRowThreads), 0, s, g);
}
void launch_threshold_symmetrize_group(hipStream_t s, const FrontItem* items, int count,
                                       double p, double mult, int binarize, int symtype,
                                       int preserve_diag, bool cut_ready, const TsDigits* digits) {
  int nmax;
  const GroupOf<FrontItem> g = front_pack(items, count, &nmax);
  if (nmax == 0) return;
  // (cut_ready: every member's cut vector is already there -- Percentile cuts come from their
  //  own per-member kernel; RowMax cuts are taken from the blur's per-strip row maxima here)
  if (!cut_ready)
    hipLaunchKernelGGL(k_cut_from_partials_g, dim3((nmax + 3) / 4, count), dim3(256), 0, s, g, p);
  const int t = (nmax + kTsTile - 1) / kTsTile;
  if (digits != nullptr) {
    GroupOf<TsDigits> dg;
    memset(&dg, 0, sizeof(dg));
    for (int z = 0; z < count; ++z) dg.s[z] = digits[z];
    hipLaunchKernelGGL(k_threshold_symmetrize_digits_g, dim3(t * (t + 1) / 2, count), dim3(256), 0,
                       s, g, dg, mult, binarize, symtype, preserve_diag);
    return;
  }
  hipLaunchKernelGGL(k_threshold_symmetrize_g, dim3(t * (t + 1) / 2, count), dim3(256), 0, s, g,
                     mult, binarize, symtype, preserve_diag);
}
void launch_cut_from_partials_group(hipStream_t s, const FrontItem* items, int count, double p) {
  int nmax;
  const GroupOf<FrontItem> g = front_pack(items, count, &nmax);
  if (nmax == 0) return;
  hipLaunchKernelGGL(k_cut_from_partials_g, dim3((nmax + 3) / 4, count), dim3(256), 0, s, g, p);
}
void launch_free_partials_reduce_group(hipStream_t s, const FreeItem* items, int count) {
  GroupOf<FreeItem> g;
  memset(&g, 0, sizeof(g));
  int nmax = 0;
  for (int z = 0; z < count; ++z) {
    g.s[z] = items[z];
    nmax = std::max(nmax, items[z].n);
  }
  if (nmax == 0) return;
  hipLaunchKernelGGL(k_free_partials_reduce_g, dim3((nmax + kTsTile - 1) / kTsTile, count),
                     dim3(256), 0, s, g);
}
void launch_scaling_vectors_group(hipStream_t s, const FrontItem* items, int count,
                                  int laplacian_type, int row_normalized) {
  int nmax;
  const GroupOf<FrontItem> g = front_pack(items, count, &nmax);
  if (nmax == 0) return;
  hipLaunchKernelGGL(k_scaling_vectors_g, dim3((nmax + 255) / 256, count), dim3(256), 0, s, g,
                     laplacian_type, row_normalized);
}
void launch_check_finite(hipStream_t s, const double* a, const double* b, int n, int* flag) {
  hipLaunchKernelGGL(k_check_finite, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, flag);
}
void launch_laplacian(hipStream_t s, const double* in, double* out, int n, int ld,
                      int laplacian_type, double* deg_ws) {
  hipLaunchKernelGGL(k_row_sum, dim3(n), dim3(kRowThreads), 0, s, in, n, ld,
                     deg_ws);
  hipLaunchKernelGGL(k_laplacian, dim3(n), dim3(kRowThreads), 0, s, in, out, n, ld,
                     laplacian_type, deg_ws);
}
void launch_symmetrize(hipStream_t s, const double* in, double* out, int n, int ld,
                       int type) {
  const int t = (n + 31) / 32;
  hipLaunchKernelGGL(k_symmetrize, dim3(t, t), dim3(256), 0, s, in, out, n, ld,
                     type);
}

}  // namespace sc
