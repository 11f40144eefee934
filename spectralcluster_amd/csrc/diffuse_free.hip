// Matrix-free Diffuse (reference refinement.py:229-245 Diffuse + RowWiseNormalize, and
// laplacian.py:41-58 on top of them) -- DESIGN.md section 3.6.
//
// For a sequence whose Diffuse is followed only by RowWiseNormalize (+ a Laplacian) the eigen
// stage never reads an ENTRY of S = A A^T.  It needs
//     S V            = A (A V)            two passes over A per Krylov block,
//     rowsum(S)      = A (A 1)            one n-vector pass,
//     rowmax(S)_i    = max_j <A_i, A_j>   the one quantity that is not a matrix-vector product.
// rowmax(S) is found without the fp64 n^3 product:
//   1. k_free_quantize   A -> 15-bit fixed point q = rint(sigma a), sigma = 32639 / max|a|, split
//                        into two signed 8-bit digits q = 256 h + l  (a single ICASSP call never
//                        runs it: its threshold + symmetrise pass writes the digits itself,
//                        rowops.hip k_threshold_symmetrize_digits + k_free_partials_reduce);
//   2. k_gemm_i8_sym     T = Q Q^T EXACTLY (integer MFMA, v_mfma_i32_32x32x32_i8: hh, hl + lh and
//                        ll products in three i32 accumulators), upper-triangle tiles, stored as
//                        fp32, row maxima M_i of T from its epilogue; the last tiles mod #CUs
//                        are cut along K and finished by k_i8_tail_finish;
//   3. k_t32_candidates  every j with
//                        T_ij >= M_i - slack_i, where slack_i is a PROVEN bound: with
//                        sigma a_ik = q_ik + d_ik, |d_ik| <= 1/2 (+ one fp64 rounding),
//                        sigma^2 S_ij - T_ij = sum_k (q_ik d_jk + d_ik q_jk + d_ik d_jk), so
//                        |sigma^2 S_ij - T_ij| <= (R_i + R_j) / 2 + n / 4,  R_i = sum_k |q_ik|;
//                        hence the true argmax j* of row i satisfies
//                        T_ij* >= M_i - (R_i + Rmax) - n / 2 - (fp32 rounding of the two T's);
//   4. k_free_row_stats  exact fp64 dot products <A_i, A_j> for the (1-3) candidates of a row:
//                        rowmax(S)_i; and rowsum(S)_i = <A_i, A 1> in the same pass over A_i.
// Rows with more candidates than kFreeCap (a handful of samples far from everything: their row
// of S is tiny against the slack) are evaluated exactly, eight at a time, as one block matvec
// S[:, rows] = A (A[rows, :]^T) (free_api.hip).
//
// Integer arithmetic makes the bound a statement about the QUANTISER alone: no assumption
// about the accumulation order or internal precision of the matrix core enters it.
// Every step also exists for a GROUP of matrices (k_*_g, blockIdx.y = member: AutoTune sweep,
// the large members of a batch group).
#include <algorithm>
#include <cstring>
#include <mutex>

#include "dpp.h"
#include "sc_internal.h"

namespace sc {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ double fr_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double fr_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// 256-thread workgroups; sm: >= 4 doubles
__device__ __forceinline__ double fr_block_max(double v, double* sm) {
  v = fr_wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
__device__ __forceinline__ double fr_block_sum(double v, double* sm) {
  v = fr_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// atomicMax on ONE word from every workgroup of a launch: 8192 same-address atomics cost the
// quantiser 19 us of 142 at n = 8192 and 30 of 60 at n = 4096 (tests/probes/quantize_probe.hip).
// A workgroup first looks at the word (agent-scope load: the value in memory, not a stale line
// of its XCD's L2) and only joins the queue if it would raise it: after the first wave of
// workgroups almost none does.
__device__ __forceinline__ void atomic_max_if_larger(unsigned long long* word, unsigned long long v) {
  const unsigned long long cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v > cur) atomicMax(word, v);
}

constexpr int kFreeOvfWords = 80;  // (free_api.hip's kOvfWords: the overflow record behind M and count)
constexpr int kPlanTotalWord = 68;  // ovf[68]: length of the tile skip list (k_free_tile_flags adds it up)

// order-preserving map float -> unsigned (for atomicMax on values of either sign)
__device__ __forceinline__ unsigned ordered_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// tile slot s of the row-major upper-triangle enumeration -> (I, J), J >= I
__device__ __forceinline__ void slot_to_tile(int slot, int nt, int* I, int* J) {
  const double b2 = 2.0 * nt + 1.0;
  int ti = (int)((b2 - sqrt(b2 * b2 - 8.0 * slot)) * 0.5);
  ti = ti < 0 ? 0 : (ti >= nt ? nt - 1 : ti);
  auto off = [&](int t) { return t * nt - (t * (t - 1)) / 2; };
  while (ti > 0 && off(ti) > slot) --ti;
  while (ti + 1 < nt && off(ti + 1) <= slot) ++ti;
  *I = ti;
  *J = ti + (slot - off(ti));
}
__device__ __forceinline__ int tile_to_slot(int I, int J, int nt) {
  return I * nt - I * (I - 1) / 2 + (J - I);
}

// ---- tile skip list ("plan") of the digit product: plan[I] = surviving tiles of tile row I,
// plan[nt + I * nt + k] = column J of its k-th survivor (k_free_tile_flags).  A workgroup that
// walks the list turns the counts into running totals in LDS (nt <= 512 ints) and finds entry
// w by bisection: no compaction kernel, no count on the host.
// (all threads of the workgroup; a barrier must follow before `pref` is read)
__device__ __forceinline__ void plan_scan(const int* __restrict__ plan, int nt, int* pref) {
  if (threadIdx.x < 64) {  // one wave: lane l owns entries [l * per, (l + 1) * per)
    const int per = (nt + 63) / 64;
    const int lane = threadIdx.x;
    int loc[8];
    int sum = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = lane * per + u;
      loc[u] = (u < per && e < nt) ? plan[e] : 0;
      sum += loc[u];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    int run = incl - sum;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = lane * per + u;
      run += loc[u];
      if (u < per && e < nt) pref[e] = run;
    }
  }
}
// entry w (< pref[nt - 1]) of the list -> (I, J)
__device__ __forceinline__ void plan_lookup(const int* __restrict__ plan, int nt, const int* pref,
                                            int w, int* I, int* J) {
  int lo = 0, hi = nt - 1;  // first I with pref[I] > w
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (pref[mid] > w) hi = mid; else lo = mid + 1;
  }
  const int before = lo > 0 ? pref[lo - 1] : 0;
  *I = lo;
  *J = plan[nt + lo * nt + (w - before)];
}
// How `count` tiles are dealt to `cus` compute units: whole rounds of one tile per workgroup,
// and -- when the last round would leave more than half of the chip idle -- its r tiles cut
// along K into f parts each (f <= 8, dividing the stage count, >= 4 stages per part, r f <= cus).
__host__ __device__ __forceinline__ void i8_split_plan(int count, int cus, int stages, int* r_out,
                                                       int* f_out) {
  *r_out = 0;
  *f_out = 1;
  const int r = count % cus;
  if (r == 0 || 2 * r > cus) return;
  const int fmax = cus / r < 8 ? cus / r : 8;
  for (int f = fmax; f >= 2; --f)
    if (stages % f == 0 && stages / f >= 4) {
      *r_out = r;
      *f_out = f;
      return;
    }
}

}  // namespace

// ---------------------------------------------------------------- scale
// scal[0] = max |a_ij| (bit pattern of a non-negative double: ordered like an unsigned integer)
__global__ __launch_bounds__(256) void k_free_absmax(const double* __restrict__ A, int n, int ld,
                                                     unsigned long long* __restrict__ amax_bits) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const double* x = A + (size_t)row * ld;
  double m = 0.0;
  for (int j = 2 * threadIdx.x; j < n; j += 512) {
    const double2 v = *reinterpret_cast<const double2*>(x + j);
    m = fmax(m, fabs(v.x));
    if (j + 1 < n) m = fmax(m, fabs(v.y));
  }
  m = fr_block_max(m, sm);
  if (threadIdx.x == 0 && m > 0.0)
    atomic_max_if_larger(amax_bits, (unsigned long long)__double_as_longlong(m));
}

// The same bound without a pass over the matrix, for A = Symmetrize(RowWiseThreshold(B)) with
// B >= 0 and a soft multiplier in [0, 1]: every entry of A is at most max_i rowmax(B)_i, and
// cut_i = rowmax(B)_i * p is what the threshold stage already holds.  `floor_value`: 1 when the
// stage writes ones (binarisation, preserved diagonal).  Any upper bound of max|a| will do for
// the quantiser (it only has to keep sigma |a| <= 32639); this one is attained.
__device__ __forceinline__ void free_amax_from_cut_body(const double* __restrict__ cut, int n,
                                                        double p, double floor_value,
                                                        double* __restrict__ scal) {
  __shared__ double sm[4];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, cut[i]);
  m = fr_block_max(m, sm);
  // (cut_i / p can round below the row maximum it came from: one part in 1e12 on top)
  if (threadIdx.x == 0) scal[0] = fmax(m / p * (1.0 + 1e-12), floor_value);
}
__global__ __launch_bounds__(256) void k_free_amax_from_cut(const double* __restrict__ cut, int n,
                                                            double p, double floor_value,
                                                            double* __restrict__ scal) {
  free_amax_from_cut_body(cut, n, p, floor_value, scal);
}
// The start of the pipeline for a GROUP of matrices (AutoTune sweep, large members of a batch
// group; blockIdx.y = member): workgroup 0 of a member turns its cut vector into max|a| and
// clears the other scalars, the others clear its words (row maxima, candidate counts, overflow
// record) -- one launch where every member had two fills and a reduction of its own.
__global__ __launch_bounds__(256) void k_free_begin_g(const GroupOf<FreeItem> g, double floor_value,
                                                      int pad_rows) {
  const FreeItem& a = g.s[blockIdx.y];
  if (a.n <= 0) return;
  if (blockIdx.x == 0) {
    free_amax_from_cut_body(a.cut, a.n, a.p, floor_value, a.scal);
    if (threadIdx.x < 3) a.scal[1 + threadIdx.x] = 0.0;
    return;
  }
  const int nwords = 2 * a.n + kFreeOvfWords;
  const int wblocks = (nwords + 1023) / 1024;
  if ((int)blockIdx.x <= wblocks) {
    const int e = ((int)blockIdx.x - 1) * 1024 + 4 * threadIdx.x;
    if (e + 3 < nwords) {
      *reinterpret_cast<int4*>(a.words + e) = make_int4(0, 0, 0, 0);
    } else {
      for (int u = e; u < nwords && u < e + 4; ++u) a.words[u] = 0;
    }
    return;
  }
  // (workgroups beyond the words, launched only when the threshold pass writes the digits: the
  //  digit rows [64 ceil(n / 64), 128 ceil(n / 128)) belong to none of its tiles -- zero)
  if (!pad_rows || a.Q == nullptr) return;
  const int t64 = (a.n + 63) / 64 * 64, t128 = (a.n + 127) / 128 * 128;
  if (t128 == t64) return;
  const size_t pitch = (size_t)2 * t64;  // bytes per digit row
  const size_t total16 = (size_t)(t128 - t64) * pitch / 16;
  int4* dst = reinterpret_cast<int4*>(a.Q + (size_t)t64 * pitch);
  const size_t stride = (size_t)(gridDim.x - 1 - wblocks) * 256;
  for (size_t u = (size_t)((int)blockIdx.x - 1 - wblocks) * 256 + threadIdx.x; u < total16; u += stride)
    dst[u] = make_int4(0, 0, 0, 0);
}

// ---------------------------------------------------------------- quantiser
// Row `row` of A -> digits.  Layout of Q: row pitch 2 Kp bytes (Kp = n rounded up to 64); the
// 64 k's of block b live in one 128-byte line: bytes [0, 64) the high digits, [64, 128) the low
// digits -- a K stage of the GEMM reads whole lines.  Rows >= n and columns >= n are zero.
// Also y1 = rowsum(A) (fp64, fixed order), R = sum |q| and its maximum.
// (PROBE, tests/probes/quantize_probe.hip only: 1 = no atomicMax, 2 = no digits, 3 = no copy-out)
template <int PROBE>
__device__ __forceinline__ void free_quantize_body(
    const double* __restrict__ A, int n, int ld, signed char* __restrict__ Q, size_t pitch, int Kp,
    double* scal, double* __restrict__ y1, double* __restrict__ R,
    unsigned long long* __restrict__ rmax_bits, double* __restrict__ q2part) {
  extern __shared__ __attribute__((aligned(16))) signed char qimg[];  // the row's digit image
  __shared__ double sm[4];
  const int row = blockIdx.x;
  signed char* qrow = Q + (size_t)row * pitch;
  if (row >= n) {
    for (int u = threadIdx.x; u < (int)(pitch / 16); u += 256)
      reinterpret_cast<int4*>(qrow)[u] = make_int4(0, 0, 0, 0);
    return;
  }
  const double amax = scal[0];
  const double sigma = (amax > 0.0 && isfinite(amax)) ? 32639.0 / amax : 0.0;
  const double* x = A + (size_t)row * ld;
  double sum = 0.0, rsum = 0.0;
  // coalesced: lane t reads the pair k = 2 t + 512 m (16 B), digits go to the LDS image.
  // FOUR pairs are requested before the first is used: with one load in flight per wave the
  // kernel ran at exactly the 4.1 TB/s that 32 KB in flight per CU and 2 us of latency give.
  for (int k0 = 2 * threadIdx.x; k0 < Kp; k0 += 2048) {
    double2 v4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + 512 * u;
      // rows are padded to ld (a multiple of 16 doubles): a pair that starts inside the row's
      // storage stays inside it
      v4[u] = (k < Kp && k < ld) ? *reinterpret_cast<const double2*>(x + k) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + 512 * u;
      if (k >= Kp) break;
      double2 v = v4[u];
      if (k >= n) v.x = 0.0;
      if (k + 1 >= n) v.y = 0.0;
      sum += v.x;
      sum += v.y;
      const double e[2] = {v.x, v.y};
      signed char hb[2], lb[2];
      double q2 = 0.0;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        double qd = rint(e[w] * sigma);
        // max|a| in scal[0] is an upper bound by construction; should a value exceed it after all,
        // the |d| <= 1/2 premise of the slack is gone: scal[3] sends the call to the explicit product
        if (fabs(qd) > 32639.0) scal[3] = 1.0;
        qd = fmin(fmax(qd, -32639.0), 32639.0);  // (NaN -> -32639: such a row is flagged later)
        const int q = (int)qd;
        const int h = (q + 128) >> 8;             // floor((q + 128) / 256): l in [-128, 127]
        const int l = q - (h << 8);
        rsum += (double)(q < 0 ? -q : q);
        q2 += qd * qd;
        hb[w] = (signed char)h;
        lb[w] = (signed char)l;
      }
      if (q2part != nullptr) {
        // the 32 lanes of a half-wave hold one 64-column block of the row (k0 = 2 t): its
        // squared digit norm, for the tile skip list ([block][row] like the fused pass)
        const double q2s = half_sum_to_last(q2);
        if ((threadIdx.x & 31) == 31) q2part[(size_t)(k >> 6) * Kp + row] = q2s;
      }
      if (PROBE == 2) continue;
      signed char* dst = qimg + (k >> 6) * 128 + (k & 63);
      *reinterpret_cast<short*>(dst) = (short)((unsigned char)hb[0] | ((unsigned short)(unsigned char)hb[1] << 8));
      *reinterpret_cast<short*>(dst + 64) = (short)((unsigned char)lb[0] | ((unsigned short)(unsigned char)lb[1] << 8));
    }
  }
  sum = fr_block_sum(sum, sm);   // (its barriers also publish the image)
  rsum = fr_block_sum(rsum, sm);
  if (PROBE != 3)
    for (int u = threadIdx.x; u < (int)(pitch / 16); u += 256)
      reinterpret_cast<int4*>(qrow)[u] = reinterpret_cast<const int4*>(qimg)[u];
  if (threadIdx.x == 0) {
    y1[row] = sum;
    R[row] = rsum;
    if (PROBE != 1) atomic_max_if_larger(rmax_bits, (unsigned long long)__double_as_longlong(rsum));
  }
}
template <int PROBE>
__global__ __launch_bounds__(256) void k_free_quantize(
    const double* __restrict__ A, int n, int ld, signed char* __restrict__ Q, size_t pitch, int Kp,
    double* scal, double* __restrict__ y1, double* __restrict__ R,
    unsigned long long* __restrict__ rmax_bits, double* __restrict__ q2part) {
  free_quantize_body<PROBE>(A, n, ld, Q, pitch, Kp, scal, y1, R, rmax_bits, q2part);
}
__global__ __launch_bounds__(256) void k_free_quantize_g(const GroupOf<FreeItem> g) {
  const FreeItem& a = g.s[blockIdx.y];
  if (a.n <= 0 || (int)blockIdx.x >= (a.n + 127) / 128 * 128) return;  // rows padded to a product tile
  const int Kp = (a.n + 63) / 64 * 64;
  free_quantize_body<0>(a.A, a.n, a.ld, a.Q, (size_t)2 * Kp, Kp, a.scal, a.y1, a.R,
                        reinterpret_cast<unsigned long long*>(a.scal) + 2, a.q2part);
}

// ---------------------------------------------------------------- tile skip list
// Most 128 x 128 tiles of T = Q Q^T cannot hold a row maximum or a candidate, and a bound that
// costs n^2 / 64 operations says which.  With the digit rows cut into 64-column segments,
//     T_ij = sum_b <q_i[b], q_j[b]>  <=  sum_b ||q_i[b]|| ||q_j[b]||          (Cauchy-Schwarz)
//          <=  B_IJ := sum_b max_{i in I} ||q_i[b]||  max_{j in J} ||q_j[b]||   for i in I, j in J,
// while row i's own diagonal entry is T_ii = ||q_i||^2 <= M_i.  The candidate threshold
// free_threshold(m, R_i, R_max, n) is monotone in m and decreasing in R_max, so
//     tau_i := free_threshold(fl(T_ii), R_i, 32639 n, n)  <=  the threshold row i ends up with,
// and a tile (I, J), I != J, with  fl_up(B_IJ) < min(min_{i in I} tau_i, min_{j in J} tau_j)
// holds only entries BELOW the final threshold of their row and of their column: none of them
// is a candidate and none is a row maximum (a threshold lies below its maximum).  Skipping the
// tile therefore changes NOTHING downstream -- M, the candidate sets and rowmax(S) are those of
// the full product, for any input (tests/test_gpu_diffuse_free.py holds both against each other
// and against the oracle).  Integer sums of q^2 are exact; the square roots are rounded up, and
// B_IJ carries 1e-12 relative + 1 absolute for its own 2 nblk roundings.
// On blob-like or speaker-turn-like inputs (clusters contiguous in time) 60-85 % of the tiles go:
// n = 8192, 8 speakers: 2080 -> ~345.  Unstructured input keeps every tile and pays one launch.

// (non-fused quantiser only; the fused threshold pass has this inside k_free_partials_reduce)
__device__ __forceinline__ void free_seg_reduce_body(const double* __restrict__ R, int n,
                                                     int nblk, const FreeSegs segs) {
  __shared__ double sq[4][64];
  const int r = threadIdx.x & 63, qd = threadIdx.x >> 6;
  const int row = blockIdx.x * 64 + r;
  const size_t rows = (size_t)64 * nblk;
  const int b0 = (int)((long long)nblk * qd / 4), b1 = (int)((long long)nblk * (qd + 1) / 4);
  double q2s = 0.0;
  for (int b = b0; b < b1; ++b) {
    const double q2 = row < n ? segs.q2part[(size_t)b * rows + row] : 0.0;
    q2s += q2;
    const double m = wave_max_to_last(q2);
    if (r == 63) segs.mx64[(size_t)blockIdx.x * nblk + b] = sqrt(m) * (1.0 + 0x1p-52);
  }
  sq[qd][r] = q2s;
  __syncthreads();
  if (qd == 0) {
    const double tii = ((sq[0][r] + sq[1][r]) + sq[2][r]) + sq[3][r];
    float t = row < n ? free_threshold((float)tii, R[row], 32639.0 * (double)n, n) : INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t = fminf(t, __shfl_xor(t, o));
    if (r == 0) segs.tau64[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void k_free_seg_reduce(const double* __restrict__ R, int n,
                                                         int nblk, const FreeSegs segs) {
  free_seg_reduce_body(R, n, nblk, segs);
}
// segment maxima + tile flags of every member of a group: workgroups [0, nblk) of a member do
// the former ... (two launches: the flags read what ALL of a member's segment workgroups wrote)
__global__ __launch_bounds__(256) void k_free_seg_reduce_g(const GroupOf<FreeItem> g) {
  const FreeItem& a = g.s[blockIdx.y];
  const int nblk = (a.n + 63) / 64;
  if (a.n <= 0 || a.plan == nullptr || (int)blockIdx.x >= nblk) return;
  free_seg_reduce_body(a.R, a.n, nblk, FreeSegs{a.q2part, a.mx64, a.tau64});
}

// One workgroup per tile row I: plan[I] = number of surviving tiles (I, J >= I), their columns in
// ascending order at plan[nt + I * nt ...].  ng = 64-row groups that exist (= nblk).
// (16 waves: tile row 0 of n = 8192 has 64 bounds of 128 terms to form, each a dependent chain of
//  loads from L2 -- with 4 waves the launch took 15 us)
constexpr int kFlagThreads = 1024;
__device__ __forceinline__ void free_tile_flags_body(const double* __restrict__ mx64,
                                                     const float* __restrict__ tau64, int nt,
                                                     int nblk, int* __restrict__ plan,
                                                     int prune, int* __restrict__ total_out) {
  __shared__ double mI[1024];       // n <= 65536: nblk <= 1024
  __shared__ unsigned char keep[512];
  const int I = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ng = nblk;
  const bool i1 = 2 * I + 1 < ng;
  for (int b = threadIdx.x; b < nblk; b += kFlagThreads)
    mI[b] = fmax(mx64[(size_t)(2 * I) * nblk + b], i1 ? mx64[(size_t)(2 * I + 1) * nblk + b] : 0.0);
  const float tauI = fminf(tau64[2 * I], i1 ? tau64[2 * I + 1] : INFINITY);
  __syncthreads();
  for (int J = I + w; J < nt; J += kFlagThreads / 64) {
    const bool j1 = 2 * J + 1 < ng;
    const double* mj0 = mx64 + (size_t)(2 * J) * nblk;
    const double* mj1 = mx64 + (size_t)(2 * J + 1) * nblk;
    double sum = 0.0;
    for (int b = lane; b < nblk; b += 64)
      sum = __builtin_fma(mI[b], fmax(mj0[b], j1 ? mj1[b] : 0.0), sum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const double bound = sum * (1.0 + 1e-12) + 1.0;
    float bf = (float)bound;
    if ((double)bf < bound) bf = nextafterf(bf, INFINITY);
    const float tau = fminf(tauI, fminf(tau64[2 * J], j1 ? tau64[2 * J + 1] : INFINITY));
    // (a NaN anywhere compares false: the tile stays)
    if (lane == 0) keep[J - I] = (J == I || !prune || !(bf < tau)) ? 1 : 0;
  }
  __syncthreads();
  if (w == 0) {
    int total = 0;
    for (int base = 0; base < nt - I; base += 64) {
      const int e = base + lane;
      const bool k = e < nt - I && keep[e] != 0;
      const unsigned long long mask = __ballot(k);
      if (k) plan[nt + I * nt + total + __popcll(mask & ((1ull << lane) - 1ull))] = I + e;
      total += __popcll(mask);
    }
    if (lane == 0) {
      plan[I] = total;
      // the list's length, for workgroups that only need to know whether they are beyond it
      // (zeroed with the handle's words before this launch)
      atomicAdd(total_out, total);
    }
  }
}
__global__ __launch_bounds__(kFlagThreads) void k_free_tile_flags_g(const GroupOf<FreeItem> g, int prune) {
  const FreeItem& a = g.s[blockIdx.y];
  const int nt = (a.n + 127) / 128;  // (kI8Tile, declared with the product below)
  if (a.n <= 0 || a.plan == nullptr || (int)blockIdx.x >= nt) return;
  free_tile_flags_body(a.mx64, a.tau64, nt, (a.n + 63) / 64, a.plan, prune,
                       a.words + 2 * (size_t)a.n + kPlanTotalWord);
}
__global__ __launch_bounds__(kFlagThreads) void k_free_tile_flags(const double* __restrict__ mx64,
                                                         const float* __restrict__ tau64, int nt,
                                                         int nblk, int* __restrict__ plan,
                                                         int prune, int* __restrict__ total_out) {
  free_tile_flags_body(mx64, tau64, nt, nblk, plan, prune, total_out);
}

// ---------------------------------------------------------------- T = Q Q^T, integer MFMA
// One workgroup (8 waves) = one 128 x 128 tile (I, J), J >= I, of T; wave (wr, wc) owns the
// 32 x 64 block at (32 wr, 64 wc): two 32 x 32 MFMA blocks, three i32 accumulators each
// (hh | hl + lh | ll), T = 65536 hh + 256 (hl + lh) + ll.
//
// K loop: stages of 64 k = two MFMA k-steps.  A stage of an operand tile is 128 rows x 128 B
// (one line per row: 64 high digits, 64 low digits) = 16 KB; A tile + B tile = 32 KB; FOUR stage
// buffers (128 KB) filled by global_load_lds_dwordx4 (LDS-DMA, no staging registers) three stages
// ahead.  Software pipeline (one s_barrier per stage, in its middle):
//     [fragments of (st, k0) already in registers]
//     DMA of stage st + 3 -> buffer (st - 1) % 4     (every wave left stage st - 1 at the last barrier)
//     4 MFMAs of (st, k0) | ds_read fragments (st, k1) | 4 MFMAs of (st, k0)
//     s_waitcnt vmcnt(4): own DMA of stage st + 2 landed;  s_barrier: everybody's did
//     4 MFMAs of (st, k1) | ds_read fragments (st + 1, k0) | 4 MFMAs of (st, k1)
// so no MFMA ever waits for an LDS read issued in its own half-stage, and a DMA has two full
// stages (~2000 cycles) to land before anyone waits for it.
// LDS image: 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 7) -- the DMA
// writes lane-linear, so the permutation is applied to the SOURCE address; with it every
// 16-lane group of a ds_read_b128 (same chunk, 16 rows) covers all 64 banks.
// Epilogue: T as fp32 into the tile's slot, and the tile's row / column maxima straight into
// M (ordered-bits atomicMax: integer, so the result does not depend on the order).
constexpr int kI8Threads = 512;
constexpr int kI8Tile = 128;
constexpr int kI8StageBytes = 32768;
constexpr int kI8Buffers = 4;

__device__ __forceinline__ void glds16(const signed char* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),
                                   (__attribute__((address_space(3))) void*)(l), 16, 0, 0);
}

struct I8Frag {
  v4i ah, al, bh[2], bl[2];
};
// split-K tail: tiles >= full_tiles of the list are computed by `parts` workgroups each, which
// leave their accumulators in ws (parts x 96 x 512 ints per tail tile) for k_i8_tail_finish
struct I8Split {
  int full_tiles, parts;
  int* ws;
  // tile skip list (k_free_tile_flags): the workgroups walk the device-built list instead of
  // `tilemap`, and derive full_tiles / parts from its length themselves (i8_split_plan over
  // `cus` compute units) -- the host never learns the count before it launches
  const int* plan = nullptr;
  int cus = 256;
  const int* total = nullptr;  // the list's length (so that surplus workgroups leave at once)
};

// PROBE (tests/probes/i8_gemm_probe.hip only; the library instantiates 0): 1 = no DMA inside
// the loop, 2 = no MFMA, 3 = no fragment reads inside the loop, 4 = no s_barrier in the loop
// (wrong results) -- what each part costs.
template <int PROBE>
__device__ __forceinline__ void gemm_i8_sym_body(
    const signed char* __restrict__ Q, size_t pitch, int nstages, const int2* __restrict__ tilemap,
    int xcd_chunk, float* __restrict__ T32, int nt, int n, unsigned* __restrict__ M,
    unsigned long long* __restrict__ probe_clk, I8Split sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // (probe only: shader cycles and 100 MHz wall ticks of this workgroup's K loop)
  const unsigned long long c_begin = probe_clk ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long t_begin = probe_clk ? wall_clock64() : 0ull;
  int tile = blockIdx.x;
  int I, J;
  if (sp.plan != nullptr) {
    // ---- device-built list: its length decides the split, the list the tile
    const int count = *sp.total;
    int r = 0, f = 1;
    if (sp.cus > 0) i8_split_plan(count, sp.cus, nstages, &r, &f);  // (0: a member of a group)
    sp.full_tiles = count - r;
    sp.parts = f;
    if (tile >= sp.full_tiles + r * f) return;  // (the grid is sized for the worst case)
    plan_scan(sp.plan, nt, reinterpret_cast<int*>(lds));
    __syncthreads();
  }
  // the tail of the tile list (what would be a last, nearly empty round of workgroups) is cut
  // along K: `parts` workgroups per tile, each with nstages / parts stages
  int part = 0, kb = 0;
  const bool split = sp.parts > 1 && tile >= sp.full_tiles;
  if (split) {
    const int tb = tile - sp.full_tiles;
    tile = sp.full_tiles + tb / sp.parts;
    part = tb % sp.parts;
    nstages /= sp.parts;
    kb = part * nstages;
  } else if (sp.plan != nullptr && sp.cus > 0) {
    // XCD x (workgroup ids go round-robin over the 8 XCDs) walks a contiguous run of the
    // row-major list: its workgroups share the digit panel of a tile row through their L2
    const int chunk = sp.full_tiles >> 3, rem = sp.full_tiles & 7, x = tile & 7;
    tile = x * chunk + (x < rem ? x : rem) + (tile >> 3);
  } else if (xcd_chunk > 0) {
    // XCD-aware order (workgroup ids go round-robin over the 8 XCDs, each with its own L2): XCD
    // x walks the contiguous run [x * xcd_chunk, (x + 1) * xcd_chunk) of the patch-ordered list
    tile = (tile & 7) * xcd_chunk + (tile >> 3);
  }
  if (sp.plan != nullptr) {
    plan_lookup(sp.plan, nt, reinterpret_cast<const int*>(lds), tile, &I, &J);
    __syncthreads();  // (the DMA is about to overwrite the running totals)
  } else {
    const int2 tij = tilemap[tile];
    I = tij.x;
    J = tij.y;
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ---- DMA sources: instruction q of wave w fills LDS units [q * 512 + w * 64, + 64) of the
  // stage image [A tile | B tile]; unit e = row * 8 + stored chunk
  const int sr = 8 * w + (lane >> 3);
  const int cx = (lane & 7) ^ ((4 * w + (lane >> 4)) & 7);
  const signed char* gA0 = Q + (size_t)(I * kI8Tile + sr) * pitch + 16 * cx + (size_t)kb * 128;
  const signed char* gA1 = gA0 + (size_t)64 * pitch;
  const signed char* gB0 = Q + (size_t)(J * kI8Tile + sr) * pitch + 16 * cx + (size_t)kb * 128;
  const signed char* gB1 = gB0 + (size_t)64 * pitch;
  auto issue = [&](int stage) {
    if (PROBE == 1 && stage > 2) return;
    unsigned char* base = lds + (stage & 3) * kI8StageBytes + w * 1024;
    const size_t off = (size_t)stage * 128;
    glds16(gA0 + off, base);
    glds16(gA1 + off, base + 8192);
    glds16(gB0 + off, base + 16384);
    glds16(gB1 + off, base + 24576);
  };
  // ---- fragment addresses: lane (rr, g) reads row rr of its 32-row block, logical chunk
  // 4 digit + 2 step + g, stored at that ^ ((rr >> 1) & 7)
  const int rr = lane & 31, g = lane >> 5;
  const int y = g ^ ((rr >> 1) & 7);
  const int wr = w >> 1, wc = w & 1;
  const int aoff = (32 * wr + rr) * 128;
  const int boff = 16384 + (64 * wc + rr) * 128;
  int coff[2][2];
#pragma unroll
  for (int dg = 0; dg < 2; ++dg)
#pragma unroll
    for (int s = 0; s < 2; ++s) coff[dg][s] = 16 * ((4 * dg + 2 * s) ^ y);
  auto read_frag = [&](I8Frag& f, int stage, int s) {
    if (PROBE == 3 && stage > 0) return;
    const unsigned char* sb = lds + (stage & 3) * kI8StageBytes;
    f.ah = *reinterpret_cast<const v4i*>(sb + aoff + coff[0][s]);
    f.al = *reinterpret_cast<const v4i*>(sb + aoff + coff[1][s]);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f.bh[cb] = *reinterpret_cast<const v4i*>(sb + boff + cb * 4096 + coff[0][s]);
      f.bl[cb] = *reinterpret_cast<const v4i*>(sb + boff + cb * 4096 + coff[1][s]);
    }
  };

  v16i hh[2], mid[2], ll[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      hh[cb][r] = 0;
      mid[cb][r] = 0;
      ll[cb][r] = 0;
    }
  }
  auto mfma_first = [&](const I8Frag& f) {  // the four products with the high digits of A
    if (PROBE == 2) {
      asm volatile("" ::"v"(f.ah), "v"(f.al), "v"(f.bh[0]), "v"(f.bh[1]), "v"(f.bl[0]), "v"(f.bl[1]));
      return;
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) hh[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.ah, f.bh[cb], hh[cb], 0, 0, 0);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) mid[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.ah, f.bl[cb], mid[cb], 0, 0, 0);
  };
  auto mfma_second = [&](const I8Frag& f) {  // ... with its low digits
    if (PROBE == 2) return;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) ll[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.al, f.bl[cb], ll[cb], 0, 0, 0);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) mid[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.al, f.bh[cb], mid[cb], 0, 0, 0);
  };

  // ---- prologue: stages 0, 1, 2 in flight; stages 0 and 1 visible; fragments of (0, k0)
  issue(0);
  if (nstages > 1) issue(1);
  if (nstages > 2) issue(2);
  if (nstages > 2)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  I8Frag f0, f1;
  read_frag(f0, 0, 0);
  for (int st = 0; st < nstages; ++st) {
    if (st + 3 < nstages) issue(st + 3);
    // (the compiler waits for EVERY outstanding LDS read before the first MFMA that needs one:
    //  the next fragments are requested behind the first four MFMAs of a half, so that wait
    //  only ever sees reads that had a full half-stage to return.  sched_barrier pins that.)
    mfma_first(f0);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(f1, st, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_second(f0);
    // own DMA of stage st + 2 landed (only stage st + 3's four may still be in flight) ...
    if (st + 3 < nstages && PROBE != 1)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and this wave's reads of stage st have returned (its buffer is refilled next round)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PROBE != 4) __builtin_amdgcn_s_barrier();
    // (the last round re-reads its own stage: straight-line code lets the compiler count the
    //  fragment reads still in flight instead of waiting for all of them before the MFMAs)
    mfma_first(f1);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(f0, st + 1 < nstages ? st + 1 : st, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_second(f1);
  }
  if (probe_clk != nullptr && threadIdx.x == 0) {
    probe_clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c_begin;
    probe_clk[2 * blockIdx.x + 1] = wall_clock64() - t_begin;
  }
  if (split) {
    // partial accumulators -> workspace (24 x 16 B per thread, each a coalesced 8 KB store of
    // the workgroup).  (A first form let the last workgroup to arrive at a ticket add the
    // others' parts in this launch: bit-equal, and 15-27 us SLOWER than no split at all -- the
    // two agent-scope fences per workgroup are an L2 write-back + invalidate of the whole XCD,
    // and 16-32 workgroups per XCD queue up behind each other for them.)
    const int tt = tile - sp.full_tiles;
    int4* ws = reinterpret_cast<int4*>(sp.ws) + (size_t)tt * sp.parts * 24 * kI8Threads + threadIdx.x;
    int4* mine = ws + (size_t)part * 24 * kI8Threads;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mine[(size_t)(cb * 4 + q) * kI8Threads] =
            make_int4(hh[cb][4 * q], hh[cb][4 * q + 1], hh[cb][4 * q + 2], hh[cb][4 * q + 3]);
        mine[(size_t)(8 + cb * 4 + q) * kI8Threads] =
            make_int4(mid[cb][4 * q], mid[cb][4 * q + 1], mid[cb][4 * q + 2], mid[cb][4 * q + 3]);
        mine[(size_t)(16 + cb * 4 + q) * kI8Threads] =
            make_int4(ll[cb][4 * q], ll[cb][4 * q + 1], ll[cb][4 * q + 2], ll[cb][4 * q + 3]);
      }
    return;  // k_i8_tail_finish adds the parts and writes the tile
  }
  // ---- epilogue: T = 65536 hh + 256 mid + ll, exact in fp64 (|T| < 2^53), stored as fp32 into
  // the tile's slot (tile-major, row-major inside).  D layout of the 32 x 32 block: lane l,
  // register r: row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
  float* out = T32 + (size_t)tile_to_slot(I, J, nt) * (kI8Tile * kI8Tile);
  float rowm[16];  // running maxima of this lane's 16 rows over its columns
  float colm[2];   // ... of its 2 columns over its 16 rows
#pragma unroll
  for (int r = 0; r < 16; ++r) rowm[r] = -INFINITY;
  const int grow0 = I * kI8Tile + 32 * wr + 4 * g;  // + (r & 3) + 8 (r >> 2)
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int col = 64 * wc + 32 * cb + rr;
    const bool col_ok = J * kI8Tile + col < n;
    colm[cb] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * g;
      const double t = (double)hh[cb][r] * 65536.0 + (double)mid[cb][r] * 256.0 + (double)ll[cb][r];
      const float tf = (float)t;
      out[row * kI8Tile + col] = tf;
      if (col_ok) rowm[r] = fmaxf(rowm[r], tf);
      if (grow0 + (r & 3) + 8 * (r >> 2) < n) colm[cb] = fmaxf(colm[cb], tf);
    }
  }
  // row maxima: over the 32 lanes of a half-wave (the columns); the DMA buffers are dead now,
  // LDS holds the per-wave partials: rows [wc][128], columns [wr][128]
  __builtin_amdgcn_s_barrier();
  float* prow = reinterpret_cast<float*>(lds);
  float* pcol = prow + 2 * 128;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = rowm[r];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if (rr == 0) prow[wc * 128 + 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * g] = v;
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const float v = fmaxf(colm[cb], __shfl_xor(colm[cb], 32));
    if (g == 0) pcol[wr * 128 + 64 * wc + 32 * cb + rr] = v;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int row = I * kI8Tile + threadIdx.x;
    const float v = fmaxf(prow[threadIdx.x], prow[128 + threadIdx.x]);
    if (row < n && v > -INFINITY) atomicMax(&M[row], ordered_bits(v));
  } else if (threadIdx.x < 256 && I != J) {
    const int c = threadIdx.x - 128;
    const int row = J * kI8Tile + c;
    const float v = fmaxf(fmaxf(pcol[c], pcol[128 + c]), fmaxf(pcol[256 + c], pcol[384 + c]));
    if (row < n && v > -INFINITY) atomicMax(&M[row], ordered_bits(v));
  }
}

template <int PROBE>
__global__ __launch_bounds__(kI8Threads) void k_gemm_i8_sym(
    const signed char* __restrict__ Q, size_t pitch, int nstages, const int2* __restrict__ tilemap,
    int xcd_chunk, float* __restrict__ T32, int nt, int n, unsigned* __restrict__ M,
    unsigned long long* __restrict__ probe_clk, I8Split sp) {
  gemm_i8_sym_body<PROBE>(Q, pitch, nstages, tilemap, xcd_chunk, T32, nt, n, M, probe_clk, sp);
}
// Grouped form (AutoTune sweep, batch_group.hip): blockIdx.y picks one of up to kGroupMax
// problems of the same size; the tiles of all of them fill the chip where one problem's 528
// (n = 4096) leave the third round of workgroups nearly empty.
// (members may differ in size -- the large members of a batch group: a product of n = 1536 is
//  78 tiles, a third of the chip, on its own)
struct I8GroupItem {
  const signed char* Q;
  float* T32;
  unsigned* M;
  const int2* tilemap;
  int n;
  const int* plan;  // the member's skip list, or nullptr
  const int* total; // ... and its length
};
__global__ __launch_bounds__(kI8Threads) void k_gemm_i8_sym_g(const GroupOf<I8GroupItem> g) {
  const I8GroupItem& a = g.s[blockIdx.y];
  const int nt = (a.n + kI8Tile - 1) / kI8Tile;
  if (a.n <= 0 || (int)blockIdx.x >= nt * (nt + 1) / 2) return;
  const int Kp = (a.n + 63) / 64 * 64;
  I8Split sp{0, 1, nullptr};
  sp.plan = a.plan;
  sp.cus = 0;
  sp.total = a.total;
  gemm_i8_sym_body<0>(a.Q, (size_t)2 * Kp, Kp / 64, a.tilemap, 0, a.T32, nt, a.n, a.M, nullptr, sp);
}

// The tail tiles' epilogue: thread t of workgroup (tt, cq) owns what thread t of the product's
// workgroup owned in accumulator registers [4 q, 4 q + 4) of column block cb (cq = 4 cb + q):
// adds the parts (integers: the order does not matter), stores the four values of T and folds
// them into the row / column maxima.
__global__ __launch_bounds__(kI8Threads) void k_i8_tail_finish(
    const int* __restrict__ wsp, int parts, int full_tiles, const int2* __restrict__ tilemap,
    float* __restrict__ T32, int nt, int n, unsigned* __restrict__ M,
    const int* __restrict__ plan, int cus, int nstages, const int* __restrict__ total) {
  const int tt = blockIdx.x, cq = blockIdx.y, cb = cq >> 2, q = cq & 3;
  int I, J;
  if (plan != nullptr) {  // (the product's own arithmetic: length of the list -> split)
    __shared__ int pref[512];
    const int count = *total;
    int r;
    i8_split_plan(count, cus, nstages, &r, &parts);
    if (parts < 2 || tt >= r) return;
    plan_scan(plan, nt, pref);
    __syncthreads();
    full_tiles = count - r;
    plan_lookup(plan, nt, pref, full_tiles + tt, &I, &J);
  } else {
    const int2 tij = tilemap[full_tiles + tt];
    I = tij.x;
    J = tij.y;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rr = lane & 31, g = lane >> 5, wr = w >> 1, wc = w & 1;
  const int4* ws = reinterpret_cast<const int4*>(wsp) + (size_t)tt * parts * 24 * kI8Threads + threadIdx.x;
  int4 a = make_int4(0, 0, 0, 0), b = a, c = a;
  for (int p = 0; p < parts; ++p) {
    const int4* src = ws + (size_t)p * 24 * kI8Threads;
    const int4 x = src[(size_t)cq * kI8Threads];
    const int4 y = src[(size_t)(8 + cq) * kI8Threads];
    const int4 z = src[(size_t)(16 + cq) * kI8Threads];
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
    c.x += z.x; c.y += z.y; c.z += z.z; c.w += z.w;
  }
  const int hh[4] = {a.x, a.y, a.z, a.w}, mid[4] = {b.x, b.y, b.z, b.w}, ll[4] = {c.x, c.y, c.z, c.w};
  float* out = T32 + (size_t)tile_to_slot(I, J, nt) * (kI8Tile * kI8Tile);
  const int col = 64 * wc + 32 * cb + rr;
  const bool col_ok = J * kI8Tile + col < n;
  float colm = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = 32 * wr + e + 8 * q + 4 * g;  // register r = 4 q + e of the product's layout
    const double t = (double)hh[e] * 65536.0 + (double)mid[e] * 256.0 + (double)ll[e];
    const float tf = (float)t;
    out[row * kI8Tile + col] = tf;
    if (I * kI8Tile + row < n) colm = fmaxf(colm, tf);
    float v = col_ok ? tf : -INFINITY;  // the row's maximum over the half-wave's 32 columns
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if (rr == 0 && I * kI8Tile + row < n && v > -INFINITY)
      atomicMax(&M[I * kI8Tile + row], ordered_bits(v));
  }
  colm = fmaxf(colm, __shfl_xor(colm, 32));
  if (g == 0 && I != J && J * kI8Tile + col < n && colm > -INFINITY)
    atomicMax(&M[J * kI8Tile + col], ordered_bits(colm));
}

// ---------------------------------------------------------------- candidates
// (M: ordered_bits() of the row maxima of T, left by the product's epilogue; 0 = nothing)
// slack of row i (in units of sigma^2 S), see the header of this file: twice the bound on
// |sigma^2 S - T| with R_j replaced by its maximum, plus the fp32 rounding of the two stored
// values that are compared (2^-24 relative each, doubled for safety)
__device__ __forceinline__ void free_append(int row, int col, int cap, int* __restrict__ count,
                                            int* __restrict__ cand) {
  const int pos = atomicAdd(&count[row], 1);
  if (pos < cap) cand[(size_t)row * cap + pos] = col;
}

// One workgroup per stored tile (I, J): every entry is looked at once, as T[row][col] against
// its row's threshold and -- off the diagonal tiles -- as T[col][row] (T is symmetric) against
// its column's.  A wave reads whole 512-byte tile rows.
__device__ __forceinline__ void t32_candidates_body(
    const float* __restrict__ T32, int nt, int n, const unsigned* __restrict__ M,
    const double* __restrict__ R, const unsigned long long* __restrict__ rmax_bits, int cap,
    int* __restrict__ count, int* __restrict__ cand, const int* __restrict__ plan) {
  __shared__ float thrI[128], thrJ[128];
  int I, J;
  size_t slot = blockIdx.x;
  if (plan != nullptr) {  // only the tiles the product computed (k_free_tile_flags)
    __shared__ int pref[512];
    const int tiles_run = count[n + kPlanTotalWord];
    if (blockIdx.x == 0 && threadIdx.x == 0) count[n + 67] = tiles_run;  // (ovf[67], for sc_diag)
    if ((int)blockIdx.x >= tiles_run) return;
    plan_scan(plan, nt, pref);
    __syncthreads();
    plan_lookup(plan, nt, pref, blockIdx.x, &I, &J);
    slot = tile_to_slot(I, J, nt);
  } else {
    slot_to_tile(blockIdx.x, nt, &I, &J);
    if (blockIdx.x == 0 && threadIdx.x == 0) count[n + 67] = nt * (nt + 1) / 2;
  }
  const float* tile = T32 + slot * (kI8Tile * kI8Tile);
  const int r0 = I * kI8Tile, c0 = J * kI8Tile;
  // the thread's 16 pieces of the tile are requested before anything else (they depend on
  // nothing): with one load in flight per thread behind the thresholds' own round trip the
  // scan ran at 3.2 TB/s
  const int c4 = 4 * (threadIdx.x & 31);
  float4 tv[16];
#pragma unroll
  for (int rg = 0; rg < 16; ++rg)
    tv[rg] = *reinterpret_cast<const float4*>(tile + (8 * rg + (threadIdx.x >> 5)) * kI8Tile + c4);
  const double Rmax = __longlong_as_double((long long)*rmax_bits);
  // (rmax_bits[1] = scal[3]: a quantiser had to clamp a finite value -- the proven slack does not
  //  hold for this matrix; the overflow count goes past anything the exact-row route accepts, so
  //  the host forms S = A A^T explicitly.  ovf = count + n.)
  if (blockIdx.x == 0 && threadIdx.x == 0 && rmax_bits[1] != 0ull) atomicMax(count + n, 1 << 20);
  if (threadIdx.x < 128) {
    const int row = r0 + threadIdx.x;
    thrI[threadIdx.x] = row < n ? free_threshold(ordered_value(M[row]), R[row], Rmax, n) : INFINITY;
  } else {
    const int row = c0 + threadIdx.x - 128;
    thrJ[threadIdx.x - 128] =
        (row < n && I != J) ? free_threshold(ordered_value(M[row]), R[row], Rmax, n) : INFINITY;
  }
  __syncthreads();
  const float tj0 = thrJ[c4], tj1 = thrJ[c4 + 1], tj2 = thrJ[c4 + 2], tj3 = thrJ[c4 + 3];
  const float tjmin = fminf(fminf(tj0, tj1), fminf(tj2, tj3));
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) {
    const int r = 8 * rg + (threadIdx.x >> 5);
    const float4 v = tv[rg];
    const float ti = thrI[r];
    const float vmax = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (vmax >= ti) {  // (rare) as entries of row r0 + r; padding columns are not entries
      const int c = c0 + c4;
      if (v.x >= ti && c < n) free_append(r0 + r, c, cap, count, cand);
      if (v.y >= ti && c + 1 < n) free_append(r0 + r, c + 1, cap, count, cand);
      if (v.z >= ti && c + 2 < n) free_append(r0 + r, c + 2, cap, count, cand);
      if (v.w >= ti && c + 3 < n) free_append(r0 + r, c + 3, cap, count, cand);
    }
    if (vmax >= tjmin && r0 + r < n) {  // as entries of rows c0 + c4 .. (thrJ = inf when unused)
      if (v.x >= tj0) free_append(c0 + c4, r0 + r, cap, count, cand);
      if (v.y >= tj1) free_append(c0 + c4 + 1, r0 + r, cap, count, cand);
      if (v.z >= tj2) free_append(c0 + c4 + 2, r0 + r, cap, count, cand);
      if (v.w >= tj3) free_append(c0 + c4 + 3, r0 + r, cap, count, cand);
    }
  }
}
__global__ __launch_bounds__(256) void k_t32_candidates(
    const float* __restrict__ T32, int nt, int n, const unsigned* __restrict__ M,
    const double* __restrict__ R, const unsigned long long* __restrict__ rmax_bits, int cap,
    int* __restrict__ count, int* __restrict__ cand, const int* __restrict__ plan) {
  t32_candidates_body(T32, nt, n, M, R, rmax_bits, cap, count, cand, plan);
}
__global__ __launch_bounds__(256) void k_t32_candidates_g(const GroupOf<FreeItem> g, int cap) {
  const FreeItem& a = g.s[blockIdx.y];
  const int nt = (a.n + kI8Tile - 1) / kI8Tile;
  if (a.n <= 0 || (int)blockIdx.x >= nt * (nt + 1) / 2) return;
  t32_candidates_body(a.T32, nt, a.n, reinterpret_cast<const unsigned*>(a.words), a.R,
                      reinterpret_cast<const unsigned long long*>(a.scal) + 2, cap, a.words + a.n,
                      a.cand, a.plan);
}


// ---------------------------------------------------------------- exact statistics of S
// rowmax(S)_i = max over the candidates j of <A_i, A_j>, rowsum(S)_i = <A_i, y1>, y1 = A 1:
// fp64, fixed order (thread t owns k = 2 t, 2 t + 1 (mod 512), then the block tree).
// ovf[0] = rows with more candidates than `cap`, ovf[1 + e] their indices (first 64),
// ovf[65] = candidates evaluated in total, ovf[66] = largest candidate count of a row.
constexpr int kFreeCapMax = 8;
// One workgroup per row; CNT candidates of this row (a compile-time count: the loads of U steps
// -- the row, y1 and the candidate rows -- are all in flight together instead of sitting behind
// one branch each).  A candidate that is the row itself (the diagonal entry <A_i, A_i>: half of
// all candidates) costs no second stream.  220-224 us at n = 8192 = 2.5 TB/s of HBM traffic
// (546 MB per launch, PMC) -- and it stays there whatever the form: round 5 measured the walk
// with 2 and with 4 steps in flight, with and without the second stream for self-candidates,
// and one WAVEFRONT per row with four rows per workgroup and shuffles instead of block
// reductions (254 us): profiles/r07c_*, r07d_*.
template <int CNT, int U>
__device__ __forceinline__ void free_row_dots(const double* __restrict__ x,
                                               const double* const* __restrict__ xj,
                                               const bool* self, const double* __restrict__ y1,
                                               int n, double* rs_out, double* acc_out) {
  double rs = 0.0;
  double acc[CNT > 0 ? CNT : 1];
#pragma unroll
  for (int c = 0; c < CNT; ++c) acc[c] = 0.0;
  for (int k0 = 2 * threadIdx.x; k0 < n; k0 += 512 * U) {
    double2 a[U], yy[U], b[U][CNT > 0 ? CNT : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 512 * u;
      const bool in = k < n;  // (rows are padded to ld, a multiple of 16 doubles)
      a[u] = in ? *reinterpret_cast<const double2*>(x + k) : make_double2(0.0, 0.0);
      yy[u] = in ? *reinterpret_cast<const double2*>(y1 + k) : make_double2(0.0, 0.0);
#pragma unroll
      for (int c = 0; c < CNT; ++c)
        b[u][c] = (in && !self[c]) ? *reinterpret_cast<const double2*>(xj[c] + k)
                                   : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 512 * u;
      if (k + 1 >= n) {
        a[u].y = 0.0;
        yy[u].y = 0.0;
#pragma unroll
        for (int c = 0; c < CNT; ++c) b[u][c].y = 0.0;
      }
      rs = __builtin_fma(a[u].x, yy[u].x, rs);
      rs = __builtin_fma(a[u].y, yy[u].y, rs);
#pragma unroll
      for (int c = 0; c < CNT; ++c) {
        const double2 bb = self[c] ? a[u] : b[u][c];
        acc[c] = __builtin_fma(a[u].x, bb.x, acc[c]);
        acc[c] = __builtin_fma(a[u].y, bb.y, acc[c]);
      }
    }
  }
  *rs_out = rs;
#pragma unroll
  for (int c = 0; c < CNT; ++c) acc_out[c] = acc[c];
}

__device__ __forceinline__ void free_row_stats_body(
    const double* __restrict__ A, int n, int ld, const double* __restrict__ y1,
    const int* __restrict__ count, const int* __restrict__ cand, int cap,
    double* __restrict__ rowmax, double* __restrict__ rowsum, int* __restrict__ ovf) {
  __shared__ double sm[4];
  const int row = blockIdx.x;
  const int total = count[row];
  const int cnt = total < cap ? total : cap;
  const double* x = A + (size_t)row * ld;
  const double* xj[kFreeCapMax];
  bool self[kFreeCapMax];
#pragma unroll
  for (int c = 0; c < kFreeCapMax; ++c) {
    const int j = c < cnt ? cand[(size_t)row * cap + c] : row;
    self[c] = j == row;
    xj[c] = A + (size_t)j * ld;
  }
  double acc[kFreeCapMax];
#pragma unroll
  for (int c = 0; c < kFreeCapMax; ++c) acc[c] = 0.0;
  double rs = 0.0;
  switch (cnt) {  // (uniform over the workgroup; 1 and 2 are nearly every row)
    case 0: free_row_dots<0, 4>(x, xj, self, y1, n, &rs, acc); break;
    case 1: free_row_dots<1, 4>(x, xj, self, y1, n, &rs, acc); break;
    case 2: free_row_dots<2, 4>(x, xj, self, y1, n, &rs, acc); break;
    case 3: free_row_dots<3, 2>(x, xj, self, y1, n, &rs, acc); break;
    case 4: free_row_dots<4, 2>(x, xj, self, y1, n, &rs, acc); break;
    default: free_row_dots<kFreeCapMax, 1>(x, xj, self, y1, n, &rs, acc); break;
  }
  rs = fr_block_sum(rs, sm);
  double best = -INFINITY;
#pragma unroll
  for (int c = 0; c < kFreeCapMax; ++c) {
    if (c < cnt) {  // (block-uniform)
      const double d = fr_block_sum(acc[c], sm);
      best = fmax(best, d);
    }
  }
  if (threadIdx.x == 0) {
    rowmax[row] = best;
    rowsum[row] = rs;
    if (total > cap) {
      const int e = atomicAdd(&ovf[0], 1);
      if (e < 64) ovf[1 + e] = row;
    }
  }
  // ovf[65] / ovf[66] (candidates evaluated, largest count): the counts are final when this
  // launch starts, so ONE workgroup adds them up.  (Every row used to send an atomicAdd and an
  // atomicMax to those two words: 16384 same-address atomics were 77 us of this kernel's 228 at
  // n = 8192, profiles/r28.)
  if (row == 0) {
    int sum = 0, big = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
      const int t = count[i];
      sum += t < cap ? t : cap;
      big = t > big ? t : big;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sum += __shfl_xor(sum, o);
      const int other = __shfl_xor(big, o);
      big = other > big ? other : big;
    }
    __shared__ int ssum[4], sbig[4];
    if ((threadIdx.x & 63) == 0) {
      ssum[threadIdx.x >> 6] = sum;
      sbig[threadIdx.x >> 6] = big;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      ovf[65] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
      ovf[66] = max(max(sbig[0], sbig[1]), max(sbig[2], sbig[3]));
    }
  }
}
__global__ __launch_bounds__(256) void k_free_row_stats(
    const double* __restrict__ A, int n, int ld, const double* __restrict__ y1,
    const int* __restrict__ count, const int* __restrict__ cand, int cap,
    double* __restrict__ rowmax, double* __restrict__ rowsum, int* __restrict__ ovf) {
  free_row_stats_body(A, n, ld, y1, count, cand, cap, rowmax, rowsum, ovf);
}
__global__ __launch_bounds__(256) void k_free_row_stats_g(const GroupOf<FreeItem> g, int cap) {
  const FreeItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.n) return;
  free_row_stats_body(a.A, a.n, a.ld, a.y1, a.words + a.n, a.cand, cap, a.rowmax, a.rowsum,
                       a.words + 2 * (size_t)a.n);
}

// ---------------------------------------------------------------- rows evaluated in full
// Vs[k][v] = A[rows[v]][k] (v < nrows, zero otherwise): the block whose product with A is
// S[:, rows] (A symmetric).  One workgroup per 32 values of k.
__global__ __launch_bounds__(256) void k_free_gather_rows(const double* __restrict__ A, int n,
                                                          int ld, const int* __restrict__ rows,
                                                          int nrows, double* __restrict__ Vs) {
  const int v = threadIdx.x & (kEigBlock - 1);
  const int k = blockIdx.x * (256 / kEigBlock) + threadIdx.x / kEigBlock;
  if (k >= n) return;
  Vs[(size_t)k * kEigBlock + v] = v < nrows ? A[(size_t)rows[v] * ld + k] : 0.0;
}
// rowmax[rows[v]] = max_k W[k][v]   (W = S[:, rows], n x kEigBlock)
__global__ __launch_bounds__(256) void k_free_colmax(const double* __restrict__ W, int n,
                                                     const int* __restrict__ rows, int nrows,
                                                     double* __restrict__ rowmax) {
  __shared__ double sm[4];
  const int v = blockIdx.x;
  if (v >= nrows) return;
  double m = -INFINITY;
  for (int k = threadIdx.x; k < n; k += 256) m = fmax(m, W[(size_t)k * kEigBlock + v]);
  m = fr_block_max(m, sm);
  if (threadIdx.x == 0) rowmax[rows[v]] = m;
}

// ---------------------------------------------------------------- launchers
int free_rows_padded(int n) { return round_up(n, kI8Tile); }
int free_k_padded(int n) { return round_up(n, 64); }
size_t free_q_bytes(int n) { return (size_t)free_rows_padded(n) * 2 * free_k_padded(n); }
size_t free_t32_bytes(int n) {
  const size_t nt = (n + kI8Tile - 1) / kI8Tile;
  return nt * (nt + 1) / 2 * kI8Tile * kI8Tile * sizeof(float);
}
int free_candidate_cap() { return kFreeCapMax; }

void launch_free_absmax(hipStream_t s, const double* A, int n, int ld, double* scal) {
  hipLaunchKernelGGL(k_free_absmax, dim3(n), dim3(256), 0, s, A, n, ld,
                     reinterpret_cast<unsigned long long*>(scal));
}

void launch_free_amax_from_cut(hipStream_t s, const double* cut, int n, double p,
                               double floor_value, double* scal) {
  hipLaunchKernelGGL(k_free_amax_from_cut, dim3(1), dim3(256), 0, s, cut, n, p, floor_value, scal);
}

void launch_free_quantize(hipStream_t s, const double* A, int n, int ld, signed char* Q,
                          double* scal, double* y1, double* R, double* q2part) {
  const int Kp = free_k_padded(n);
  SC_OPT_IN_LDS(k_free_quantize<0>, 2 * 65536);  // (n <= 65536: the row image is 2 Kp bytes)
  hipLaunchKernelGGL(k_free_quantize<0>, dim3(free_rows_padded(n)), dim3(256), (size_t)2 * Kp, s, A,
                     n, ld, Q, (size_t)2 * Kp, Kp, scal, y1, R,
                     reinterpret_cast<unsigned long long*>(scal) + 2, q2part);
}
// ---- tile skip list: sizes, and the two small launches that build it
size_t free_q2part_bytes(int n) { return (size_t)free_k_padded(n) * (free_k_padded(n) / 64) * sizeof(double); }
size_t free_mx64_bytes(int n) { return (size_t)(free_k_padded(n) / 64) * (free_k_padded(n) / 64) * sizeof(double); }
size_t free_tau64_bytes(int n) { return (size_t)(free_k_padded(n) / 64) * sizeof(float); }
size_t free_plan_bytes(int n) {
  const size_t nt = (n + kI8Tile - 1) / kI8Tile;
  return (nt + nt * nt) * sizeof(int);
}
void launch_free_seg_reduce(hipStream_t s, const double* R, int n, const FreeSegs& segs) {
  const int nblk = free_k_padded(n) / 64;
  hipLaunchKernelGGL(k_free_seg_reduce, dim3(nblk), dim3(256), 0, s, R, n, nblk, segs);
}
void launch_free_tile_flags(hipStream_t s, const double* mx64, const float* tau64, int n,
                            int* plan, bool prune, int* words) {
  const int nt = (n + kI8Tile - 1) / kI8Tile;
  hipLaunchKernelGGL(k_free_tile_flags, dim3(nt), dim3(kFlagThreads), 0, s, mx64, tau64, nt,
                     free_k_padded(n) / 64, plan, prune ? 1 : 0,
                     words + 2 * (size_t)n + kPlanTotalWord);
}

// The tail of the product: `tiles` tiles on `cus` compute units leave tiles % cus tiles for a
// last round in which most of the chip idles (n = 8192: 2080 = 8 x 256 + 32).  Those tiles are
// cut along K into the largest number of parts (<= 8, dividing the stage count, >= 4 stages
// each) that still fits one round.  parts = 1: no tail worth cutting.
static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    else
      cus = 256;
  }
  return cus;
}
void free_i8_split_plan(int n, int* tail_tiles, int* parts) {
  const int nt = (n + kI8Tile - 1) / kI8Tile;
  const int tiles = nt * (nt + 1) / 2;
  const int stages = free_k_padded(n) / 64;
  const int cus = device_cus();
  *tail_tiles = 0;
  *parts = 1;
  const int r = tiles % cus;
  if (tiles <= cus || r == 0 || 2 * r > cus) return;
  for (int f = std::min(8, cus / r); f >= 2; --f)
    if (stages % f == 0 && stages / f >= 4) {
      *tail_tiles = r;
      *parts = f;
      return;
    }
}
size_t free_i8_split_bytes(int n) {
  int r, f;
  free_i8_split_plan(n, &r, &f);
  return f > 1 ? (size_t)r * f * 96 * kI8Threads * sizeof(int) : 0;
}
// (with a device-built skip list the number of tiles is not known on the host: the workspace
//  holds the worst case, r f <= #CUs partial tiles)
size_t free_i8_split_bytes_plan() { return (size_t)device_cus() * 96 * kI8Threads * sizeof(int); }

void launch_gemm_i8_sym(hipStream_t s, const signed char* Q, int n, const int2* tilemap,
                        float* T32, unsigned* M, int* split_ws, const int* plan) {
  const int nt = (n + kI8Tile - 1) / kI8Tile;
  const int tiles = nt * (nt + 1) / 2;
  const int Kp = free_k_padded(n);
  const int lds = kI8Buffers * kI8StageBytes;
  if (plan != nullptr && split_ws != nullptr) {
    // every workgroup reads the list's length and deals the tiles itself (i8_split_plan); the
    // grid covers the longest list + one round of K parts, the surplus exits at once
    const int cus = device_cus();
    I8Split sp{0, 1, split_ws};
    sp.plan = plan;
    sp.cus = cus;
    sp.total = reinterpret_cast<const int*>(M) + 2 * (size_t)n + kPlanTotalWord;
    SC_OPT_IN_LDS(k_gemm_i8_sym<0>, lds);
    hipLaunchKernelGGL(k_gemm_i8_sym<0>, dim3(tiles + cus), dim3(kI8Threads), lds, s, Q,
                       (size_t)2 * Kp, Kp / 64, tilemap, 0, T32, nt, n, M,
                       static_cast<unsigned long long*>(nullptr), sp);
    hipLaunchKernelGGL(k_i8_tail_finish, dim3(cus / 2, 8), dim3(kI8Threads), 0, s, split_ws, 1, 0,
                       tilemap, T32, nt, n, M, plan, cus, Kp / 64, sp.total);
    return;
  }
  int r = 0, f = 1;
  if (split_ws != nullptr) free_i8_split_plan(n, &r, &f);
  const int full = tiles - r;
  const int xcd_chunk = (full % 8 == 0 && full >= 512) ? full / 8 : 0;
  SC_OPT_IN_LDS(k_gemm_i8_sym<0>, lds);
  hipLaunchKernelGGL(k_gemm_i8_sym<0>, dim3(full + r * f), dim3(kI8Threads), lds, s, Q,
                     (size_t)2 * Kp, Kp / 64, tilemap, xcd_chunk, T32, nt, n, M,
                     static_cast<unsigned long long*>(nullptr), I8Split{full, f, split_ws});
  if (f > 1)
    hipLaunchKernelGGL(k_i8_tail_finish, dim3(r, 8), dim3(kI8Threads), 0, s, split_ws, f, full,
                       tilemap, T32, nt, n, M, static_cast<const int*>(nullptr), 0, 0,
                       static_cast<const int*>(nullptr));
}

void launch_gemm_i8_sym_group(hipStream_t s, const signed char* const* Q, float* const* T32,
                              unsigned* const* M, int count, const int* ns,
                              const int2* const* tilemaps, const int* const* plans) {
  const int lds = kI8Buffers * kI8StageBytes;
  SC_OPT_IN_LDS(k_gemm_i8_sym_g, lds);
  GroupOf<I8GroupItem> g;
  memset(&g, 0, sizeof(g));
  int tiles = 0;
  for (int z = 0; z < count; ++z) {
    // (M = the member's words: M | count | ovf)
    g.s[z] = I8GroupItem{Q[z], T32[z], M[z], tilemaps[z], ns[z], plans ? plans[z] : nullptr,
                         reinterpret_cast<const int*>(M[z]) + 2 * (size_t)ns[z] + kPlanTotalWord};
    const int nt = (ns[z] + kI8Tile - 1) / kI8Tile;
    tiles = std::max(tiles, nt * (nt + 1) / 2);
  }
  if (tiles == 0) return;
  hipLaunchKernelGGL(k_gemm_i8_sym_g, dim3(tiles, count), dim3(kI8Threads), lds, s, g);
}

void launch_t32_candidates(hipStream_t s, const float* T32, int n, const unsigned* M,
                           const double* R, const double* scal, int* count, int* cand,
                           const int* plan) {
  const int nt = (n + kI8Tile - 1) / kI8Tile;
  hipLaunchKernelGGL(k_t32_candidates, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, T32, nt, n, M, R,
                     reinterpret_cast<const unsigned long long*>(scal) + 2, kFreeCapMax, count,
                     cand, plan);
}

void launch_free_row_stats(hipStream_t s, const double* A, int n, int ld, const double* y1,
                           const int* count, const int* cand, double* rowmax, double* rowsum,
                           int* ovf) {
  hipLaunchKernelGGL(k_free_row_stats, dim3(n), dim3(256), 0, s, A, n, ld, y1, count, cand,
                     kFreeCapMax, rowmax, rowsum, ovf);
}

// ---- the same steps for a group of matrices (FreeItem per member, n = 0: idle)
static GroupOf<FreeItem> free_pack(const FreeItem* items, int count, int* nmax) {
  GroupOf<FreeItem> g;
  memset(&g, 0, sizeof(g));
  *nmax = 0;
  for (int z = 0; z < count; ++z) {
    g.s[z] = items[z];
    *nmax = std::max(*nmax, items[z].n);
  }
  return g;
}
void launch_free_begin_group(hipStream_t s, const FreeItem* items, int count, double floor_value,
                             bool pad_rows) {
  int nmax;
  const GroupOf<FreeItem> g = free_pack(items, count, &nmax);
  if (nmax == 0) return;
  const int nwords = 2 * nmax + kFreeOvfWords;
  // (pad_rows: 32 more workgroups per member clear the digit rows no threshold tile writes)
  hipLaunchKernelGGL(k_free_begin_g, dim3(1 + (nwords + 1023) / 1024 + (pad_rows ? 32 : 0), count),
                     dim3(256), 0, s, g, floor_value, pad_rows ? 1 : 0);
}
void launch_free_quantize_group(hipStream_t s, const FreeItem* items, int count) {
  int nmax;
  const GroupOf<FreeItem> g = free_pack(items, count, &nmax);
  if (nmax == 0) return;
  SC_OPT_IN_LDS(k_free_quantize_g, 2 * 65536);
  hipLaunchKernelGGL(k_free_quantize_g, dim3(free_rows_padded(nmax), count), dim3(256),
                     (size_t)2 * free_k_padded(nmax), s, g);
}
void launch_free_tile_flags_group(hipStream_t s, const FreeItem* items, int count, bool prune,
                                  bool seg_reduce) {
  int nmax;
  const GroupOf<FreeItem> g = free_pack(items, count, &nmax);
  if (nmax == 0) return;
  if (seg_reduce)
    hipLaunchKernelGGL(k_free_seg_reduce_g, dim3(free_k_padded(nmax) / 64, count), dim3(256), 0, s, g);
  hipLaunchKernelGGL(k_free_tile_flags_g, dim3((nmax + kI8Tile - 1) / kI8Tile, count),
                     dim3(kFlagThreads), 0, s, g, prune ? 1 : 0);
}
void launch_free_scan_stats_group(hipStream_t s, const FreeItem* items, int count) {
  int nmax;
  const GroupOf<FreeItem> g = free_pack(items, count, &nmax);
  if (nmax == 0) return;
  const int nt = (nmax + kI8Tile - 1) / kI8Tile;
  hipLaunchKernelGGL(k_t32_candidates_g, dim3(nt * (nt + 1) / 2, count), dim3(256), 0, s, g,
                     kFreeCapMax);
  hipLaunchKernelGGL(k_free_row_stats_g, dim3(nmax, count), dim3(256), 0, s, g, kFreeCapMax);
}

void launch_free_gather_rows(hipStream_t s, const double* A, int n, int ld, const int* rows,
                             int nrows, double* Vs) {
  const int per = 256 / kEigBlock;
  hipLaunchKernelGGL(k_free_gather_rows, dim3((n + per - 1) / per), dim3(256), 0, s, A, n, ld,
                     rows, nrows, Vs);
}

void launch_free_colmax(hipStream_t s, const double* W, int n, const int* rows, int nrows,
                        double* rowmax) {
  hipLaunchKernelGGL(k_free_colmax, dim3(nrows), dim3(256), 0, s, W, n, rows, nrows, rowmax);
}

}  // namespace sc
