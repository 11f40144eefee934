// fp64 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T  (both operands
// row-major with K contiguous, i.e. the "NT" form both hot products have):
//   * affinity  C = Xn Xn^T, epilogue (c + 1) / 2      (reference utils.py:35-39)
//   * Diffuse   C = A A^T                               (reference refinement.py:234)
// Both outputs are symmetric, so the SYM variant computes only tile pairs
// (ti <= tj) and writes the mirror tile transposed: half the flops, and the
// result is exactly symmetric (what numpy's syrk-backed A @ A.T gives).
//
// Tiling: 128x128 block tile, BK = 16, 256 threads = 4 waves, each wave a 64x64
// sub-tile = 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 acc VGPRs), two workgroups
// per CU.  LDS tiles are [128][16] doubles; the 16-byte chunk kc of row r sits at chunk
// position kc ^ ((r >> 1) & 7) (lds_chunk_off), which makes both the ds_write_b128 of the
// staging step and the ds_read_b128 fragment reads conflict-free.  Two LDS buffers and two
// register sets of fragments: see the main loop of k_gemm_nt.
//
// Measured (SC_GEMM_CLOCK=1 probe, n = 8192): 4.51 M shader cycles per tile against an
// MFMA-bound minimum of 4.19 M (93 %); the shader clock the power management sustains
// under this kernel is 2.35 GHz on constant data and 2.0-2.1 GHz on random dense data
// (2.4 GHz nominal) -- the kernel is power-limited, not issue-limited.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include <cstring>

#include "sc_internal.h"

namespace sc {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 16;

// 16-byte chunk kc (k = 2kc, 2kc+1) of row `row` sits at chunk position kc ^ ((row >> 1) & 7)
// of the row's 128 bytes: the 16 lanes of a ds_read_b128 group (rows li = 0..15, one kc)
// then cover 16 distinct 16-byte slots of a 256-byte bank window -- conflict-free.
__device__ __forceinline__ int lds_chunk_off(int row, int kc) {
  return row * BK + ((2 * kc) ^ (row & 14));
}

// Maps a linear tile id to (ti, tj): row by row over the upper triangle when SYM,
// plain row-major otherwise.
template <bool SYM>
__device__ __forceinline__ void tile_coords(int id, int ntiles_m, int ntiles_n,
                                            int* ti, int* tj) {
  if (SYM) {
    int r = 0, rowlen = ntiles_m;
    while (id >= rowlen) {
      id -= rowlen;
      --rowlen;
      ++r;
    }
    *ti = r;
    *tj = r + id;
  } else {
    *ti = id / ntiles_n;
    *tj = id - (*ti) * ntiles_n;
  }
}

// Optional fused row statistics of the (symmetric) result, so the next stage does not
// have to re-read the n x n matrix:
//   mode 1 (Diffuse)  : per row  max_j C_ij  and  sum_j C_ij
//   mode 2 (affinity) : per row  max_{j != i} C_ij      (CropDiagonal's value, pre-clamp)
// A tile (ti, tj) contributes, for each of its 128 rows, the max / sum over its 128 columns
// into slot tj of that row, and -- being the mirror of tile (tj, ti) -- for each of its
// columns the max / sum over its rows into slot ti of row `col`.  Every (row, slot) of the
// n x ntiles partial arrays is written exactly once; k_gemm_stats_reduce finishes in a fixed
// order (deterministic).  `scratch` is >= 1024 doubles of LDS (the operand tiles are dead).
struct GemmStats {
  double* pmax;   // n x ntiles
  double* psum;   // n x ntiles (mode 1 only)
  int mode;       // 0 = off
  const double* addend;  // kEpiAdd: C = A B^T + addend (same ld as C; may not alias C)
};

template <int EPI, bool SYM>
__device__ __forceinline__ void tile_row_stats(const v4f64 (&acc)[4][4], int ti, int tj,
                                               int ntiles, int M, int N, int tid,
                                               const GemmStats& st, double* scratch) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  double* rmax = scratch;            // [2][128]  (wc, row)
  double* rsum = scratch + 256;      // [2][128]
  double* cmax = scratch + 512;      // [2][128]  (wr, col)
  double* csum = scratch + 768;      // [2][128]
  const bool diag_tile = ti == tj;
  // --- per-row partials over this wave's 64 columns
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = wr * 64 + m * 16 + lg + 4 * r;
      double mx = -INFINITY, sm = 0.0;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        const int lcol = wc * 64 + nn * 16 + li;
        const bool inside = tj * BN + lcol < N;
        const bool skip = st.mode == 2 && diag_tile && lrow == lcol;
        double x = acc[m][nn][r];
        if (EPI == kEpiAffinity) x = __builtin_fma(x, 0.5, 0.5);
        if (inside && !skip) mx = fmax(mx, x);
        if (inside) sm += x;
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        mx = fmax(mx, __shfl_xor(mx, o));
        sm += __shfl_xor(sm, o);
      }
      if (li == 0) {
        rmax[wc * 128 + lrow] = mx;
        rsum[wc * 128 + lrow] = sm;
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the live range of each reduction short
    }
  }
  // --- per-column partials over this wave's 64 rows (the mirror tile's rows)
  if (SYM && !diag_tile) {
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      const int lcol = wc * 64 + nn * 16 + li;
      double mx = -INFINITY, sm = 0.0;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lrow = wr * 64 + m * 16 + lg + 4 * r;
          double x = acc[m][nn][r];
          if (EPI == kEpiAffinity) x = __builtin_fma(x, 0.5, 0.5);
          if (ti * BM + lrow < M) {
            mx = fmax(mx, x);
            sm += x;
          }
        }
      mx = fmax(mx, __shfl_xor(mx, 16));
      sm += __shfl_xor(sm, 16);
      mx = fmax(mx, __shfl_xor(mx, 32));
      sm += __shfl_xor(sm, 32);
      if (lg == 0) {
        cmax[wr * 128 + lcol] = mx;
        csum[wr * 128 + lcol] = sm;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  if (tid < 128) {
    const int row = ti * BM + tid;
    if (row < M) {
      st.pmax[(size_t)row * ntiles + tj] = fmax(rmax[tid], rmax[128 + tid]);
      if (st.mode == 1) st.psum[(size_t)row * ntiles + tj] = rsum[tid] + rsum[128 + tid];
    }
  } else if (SYM && !diag_tile) {
    const int c = tid - 128;
    const int row = tj * BN + c;
    if (row < N) {
      st.pmax[(size_t)row * ntiles + ti] = fmax(cmax[c], cmax[128 + c]);
      if (st.mode == 1) st.psum[(size_t)row * ntiles + ti] = csum[c] + csum[128 + c];
    }
  }
}

// v_max_f64 without the canonicalising self-max hipcc puts in front of every fmax whose
// operand it cannot prove quiet (an MFMA result, a value read back from LDS): the epilogue is
// issue-bound, and those were a quarter of its vector instructions.
__device__ __forceinline__ double vmax64(double a, double b) {
  double d;
  asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// LDS map of an epilogue (doubles; the 8192 of the operand tiles, dead by then):
//   [0, 1024)     row partials per wave column half / finished column partials (tile_row_stats)
//   [1024, 3072)  column partials of the full-tile statistics  [2 stats][wr * 4 + lg][128]
//   [3072, 8192)  per wave: the transposed row partials of the full-tile statistics, then the
//                 transposed staging of the mirror tile (1280 per wave)
constexpr int kEpiColPart = 1024;
constexpr int kEpiStage = 3072;
constexpr int kEpiStageWave = 1280;
constexpr int kEpiRowPitch = 17;  // 64 rows x 16 partials per wave, odd pitch: conflict-free

// The statistics of tile_row_stats for a tile that lies wholly inside the matrix (all of them
// when n is a multiple of 128, all but the last tile row / column otherwise).  Same values
// (max is exact; the sums are taken in a different, equally fixed order), a third of the
// instructions: no per-element guards, and the 16-lane reductions of the row partials go
// through a transposed copy in LDS (16 writes + 16 reads + 15 operations per statistic) in
// place of 64 butterfly exchanges of two ds_bpermute each.
template <int EPI, bool SYM>
__device__ __forceinline__ void tile_row_stats_full(const v4f64 (&acc)[4][4], int ti, int tj,
                                                    int ntiles, int tid, const GemmStats& st,
                                                    double* smem) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const bool diag_tile = ti == tj;
  const bool sums = st.mode == 1;
  // CropDiagonal's value skips the diagonal: it runs through the (m, m) accumulators of the
  // two waves on the tile's diagonal, at lane-local row == column
  const bool crop = st.mode == 2 && diag_tile && wr == wc;
  double* rmax = smem;        // [2][128]  (wc, row)
  double* rsum = smem + 256;  // [2][128]
  double* cpart = smem + kEpiColPart;
  double* T = smem + kEpiStage + wave * (64 * kEpiRowPitch);
  // (the affinity epilogue (v + 1) / 2 is applied at each use: done once in place, hipcc keeps
  //  the raw and the finished accumulators live side by side and spills 240 registers)
  auto el = [&](int m, int nn, int r) -> double {
    const double x = acc[m][nn][r];
    return EPI == kEpiAffinity ? __builtin_fma(x, 0.5, 0.5) : x;
  };
  // --- rows: over the 4 column blocks in the lane, then over the 16 lanes through T
  auto rows_of = [&](auto is_sum) {
    constexpr bool SUM = decltype(is_sum)::value;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double a[4];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) a[nn] = el(m, nn, r);
        double v;
        if (SUM) {
          v = (a[0] + a[1]) + (a[2] + a[3]);
        } else {
          if (crop && lg + 4 * r == li) a[m] = -INFINITY;
          v = vmax64(vmax64(a[0], a[1]), vmax64(a[2], a[3]));
        }
        T[(m * 16 + lg + 4 * r) * kEpiRowPitch + li] = v;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double t[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] = T[lane * kEpiRowPitch + j];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int j = 0; j < w; ++j) t[j] = SUM ? t[j] + t[j + w] : vmax64(t[j], t[j + w]);
    (SUM ? rsum : rmax)[wc * 128 + wr * 64 + lane] = t[0];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  rows_of(std::false_type{});
  if (sums) rows_of(std::true_type{});
  // --- columns (the mirror tile's rows): over the 16 rows in the lane; the 4 lane groups and
  // the 2 wave rows are combined by the finishing threads below
  const bool mirror = SYM && !diag_tile;
  if (mirror) {
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      double mx[4], sm[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double x0 = el(m, nn, 0), x1 = el(m, nn, 1), x2 = el(m, nn, 2), x3 = el(m, nn, 3);
        mx[m] = vmax64(vmax64(x0, x1), vmax64(x2, x3));
        sm[m] = (x0 + x1) + (x2 + x3);
      }
      const int slot = (wr * 4 + lg) * 128 + wc * 64 + nn * 16 + li;
      cpart[slot] = vmax64(vmax64(mx[0], mx[1]), vmax64(mx[2], mx[3]));
      if (sums) cpart[1024 + slot] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
  }
  __syncthreads();
  if (tid < 128) {
    const size_t at = (size_t)(ti * BM + tid) * ntiles + tj;
    st.pmax[at] = vmax64(rmax[tid], rmax[128 + tid]);
    if (sums) st.psum[at] = rsum[tid] + rsum[128 + tid];
  } else if (mirror) {
    const int c = tid - 128;
    const size_t at = (size_t)(tj * BN + c) * ntiles + ti;
    double t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = cpart[j * 128 + c];
    st.pmax[at] = vmax64(vmax64(vmax64(t[0], t[1]), vmax64(t[2], t[3])),
                         vmax64(vmax64(t[4], t[5]), vmax64(t[6], t[7])));
    if (sums) {
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = cpart[1024 + j * 128 + c];
      st.psum[at] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
  }
}

// rowmax[i] / rowsum[i] from the per-tile partials (fixed slot order)
__global__ void k_gemm_stats_reduce(const double* __restrict__ pmax,
                                    const double* __restrict__ psum, int n, int ntiles,
                                    int mode, double* __restrict__ rowmax,
                                    double* __restrict__ rowsum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double mx = -INFINITY, sm = 0.0;
  for (int t = 0; t < ntiles; ++t) {
    mx = fmax(mx, pmax[(size_t)i * ntiles + t]);
    if (mode == 1) sm += psum[(size_t)i * ntiles + t];
  }
  if (mode == 2) mx = fmax(mx, 0.0);  // CropDiagonal: the zero-filled diagonal takes part
  rowmax[i] = mx;
  if (mode == 1) rowsum[i] = sm;
}

// The same sums in the same order, with coalesced loads: a workgroup stages the partials of 32
// rows (32 x ntiles each, contiguous) through LDS, then one thread per row adds them in slot
// order.  (The one-thread-per-row form above reads 64 consecutive doubles per thread: 20 us at
// n = 8192 where this one takes a quarter of that.)  LDS: 2 x 32 x (ntiles + 1) doubles.
constexpr int kStatRows = 32;
constexpr int kStatMaxTiles = 120;  // 2 x 32 x (ntiles + 1) doubles within the 64 KB default
__device__ __forceinline__ void gemm_stats_reduce_lds_body(
    const double* __restrict__ pmax, const double* __restrict__ psum, int n, int ntiles, int mode,
    double* __restrict__ rowmax, double* __restrict__ rowsum) {
  extern __shared__ __attribute__((aligned(16))) double st_smem[];
  const int pitch = ntiles + 1;
  double* lmax = st_smem;
  double* lsum = st_smem + kStatRows * pitch;
  const int r0 = blockIdx.x * kStatRows;
  const int rows = min(kStatRows, n - r0);
  const int total = rows * ntiles;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int rr = e / ntiles, t = e - rr * ntiles;
    lmax[rr * pitch + t] = pmax[(size_t)r0 * ntiles + e];
    if (mode == 1) lsum[rr * pitch + t] = psum[(size_t)r0 * ntiles + e];
  }
  __syncthreads();
  if ((int)threadIdx.x >= rows) return;
  double mx = -INFINITY, sm = 0.0;
  for (int t = 0; t < ntiles; ++t) {
    mx = fmax(mx, lmax[threadIdx.x * pitch + t]);
    if (mode == 1) sm += lsum[threadIdx.x * pitch + t];
  }
  if (mode == 2) mx = fmax(mx, 0.0);  // CropDiagonal: the zero-filled diagonal takes part
  rowmax[r0 + threadIdx.x] = mx;
  if (mode == 1) rowsum[r0 + threadIdx.x] = sm;
}
__global__ __launch_bounds__(256) void k_gemm_stats_reduce_lds(
    const double* __restrict__ pmax, const double* __restrict__ psum, int n, int ntiles, int mode,
    double* __restrict__ rowmax, double* __restrict__ rowsum) {
  gemm_stats_reduce_lds_body(pmax, psum, n, ntiles, mode, rowmax, rowsum);
}

// One workgroup = one 128x128 output tile, or one K chunk of one.  One launch holds two kinds of work units.  Workgroups [0, full_tiles) each compute a whole
// tile: full K, epilogue + store to C (and the mirror tile when SYM).  The tiles left over
// after the last full wave of workgroups are split over K into `ksplit` chunks each: the
// workgroups after full_tiles write raw accumulators to `partial` (fragment order), to be
// summed by k_gemm_reduce.  Being last in dispatch order they fill the slots that free up
// while the last whole tiles drain, so the chip does not idle on a ragged tail.
// Grouped form (GROUPED, k_gemm_nt_g): one launch computes the products A_z A_z^T of up to
// kGroupMax independent symmetric problems (the members of a batch group, batch_group.hip).
// Workgroup b takes whole tile b - first[z] of the member z whose run [first[z], first[z+1])
// holds b: a short utterance's 36..300 tiles cannot fill the chip (and pay a split over K with
// partial stores and a reduce for trying), the tiles of 16 of them can.  Everything after the
// lookup is the single-problem tile body with the member's operands.
struct GemmMember {
  const double* A;
  double* C;
  double* pmax;
  double* psum;
  const int2* tilemap;
  int lda, ldc, n, K, nt;
};
struct GemmGroup {
  GemmMember m[kGroupMax];
  int first[kGroupMax + 1];
};

// prologue / epilogue at the top issue priority (bit 0) and non-temporal C stores (bit 1):
// 7.85 -> 7.78-7.83 ms on the Diffuse product (profiles/r02)
constexpr int kEdgePrio = 3;

template <int EPI, bool SYM, bool GROUPED>
__device__ __forceinline__ void gemm_nt_body(const double* __restrict__ A,
                                                 int lda,
                                                 const double* __restrict__ B,
                                                 int ldb, double* __restrict__ C,
                                                 int ldc, int M, int N, int K,
                                                 int ntiles_m, int ntiles_n,
                                                 int full_tiles, int ksplit_tail,
                                                 double* __restrict__ partial,
                                                 double* __restrict__ probe_out,
                                                 const int2* __restrict__ tilemap,
                                                 int xcd_chunk, GemmStats stats,
                                                 int* __restrict__ queue,
                                                 int nunits, int persist,
                                                 const GemmGroup* __restrict__ grp) {
  // one 64 KB block: As[2] | Bs[2] in the K loop, reduction scratch + the transposed
  // staging of the mirror tile in the epilogue
  __shared__ __attribute__((aligned(16))) double smem[2 * BM * BK + 2 * BN * BK];
  double(*As)[BM * BK] = reinterpret_cast<double(*)[BM * BK]>(smem);
  double(*Bs)[BN * BK] = reinterpret_cast<double(*)[BN * BK]>(smem + 2 * BM * BK);

  // Prologue and epilogue at the TOP issue priority: the partner workgroup of this CU is in
  // its K loop at priority 1-2, and at priority 0 the ~2000 scalar / vector / store
  // instructions of an epilogue only get the issue slots its MFMA stream leaves -- the tile
  // timeline shows 86 us between one tile's K loop and the next one's on the same slot,
  // during which this half of the CU's MFMA capacity is lost (one workgroup cannot use
  // more than its own share: staggering the partners did not help, profiles/r02_i).
  // Persistent form (persist != 0, with a queue): the launch has one workgroup per resident
  // slot and every workgroup keeps drawing items until none is left -- no teardown / dispatch
  // between tiles (~30 us of the 86 us gap on a slot), and the split-K units of the leftover
  // tiles are taken by whoever finishes first.
  for (int iter = 0;; ++iter) {
  if (iter > 0) {
    if (GROUPED || !persist || queue == nullptr) break;
    __syncthreads();  // the previous item's epilogue is done with the LDS
  }
  if (kEdgePrio & 1) __builtin_amdgcn_s_setprio(3);
  int ti, tj;
  // Work item of this workgroup.  Static (queue == nullptr): by block index.  Persistent
  // (queue): one workgroup per resident slot, each DRAWS its items -- whole tiles from its
  // XCD's run of the tile list, in order, then the split-K units of the leftover tiles from
  // one global counter, then other XCDs' tiles.  Which workgroup computes which item does
  // not change any result.  (Round 2 also tried shifting the two workgroups of a CU against
  // each other by a split unit, and a K-window throttle that kept an XCD's tiles in one L2
  // window: both measured slower, DESIGN_HISTORY.md section 3.3; the code is gone.)
  int item_blk = (int)blockIdx.x;
  if constexpr (GROUPED) {
    // XCD-aware order (workgroup ids go round-robin over the 8 XCDs, each with its own L2):
    // XCD x walks the contiguous run [x * xcd_chunk, (x + 1) * xcd_chunk) of the concatenated,
    // patch-ordered tile lists, so the ~64 tiles it has in flight come from one 8 x 8 patch of
    // one member and share 8 + 8 operand panels (by block index they would share 1 + 8)
    item_blk = (item_blk & 7) * xcd_chunk + (item_blk >> 3);
    if (item_blk >= grp->first[kGroupMax]) continue;  // (the runs' ragged end)
    int z = 0;
    while (item_blk >= grp->first[z + 1]) ++z;  // (uniform: scalar loads from the kernel arguments)
    const GemmMember& g = grp->m[z];
    item_blk -= grp->first[z];
    A = B = g.A;
    lda = ldb = g.lda;
    C = g.C;
    ldc = g.ldc;
    M = N = g.n;
    K = g.K;
    ntiles_m = ntiles_n = g.nt;
    tilemap = g.tilemap;
    stats.pmax = g.pmax;
    stats.psum = g.psum;
  }
  if (queue != nullptr) {
    int* s_item = reinterpret_cast<int*>(smem);  // (one LDS object per kernel: no second array)
    if (threadIdx.x == 0) {
      const int x = (int)blockIdx.x & 7;
      auto draw = [&](int* counter, int limit) -> int {
        if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= limit)
          return -1;
        const int v = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        return v < limit ? v : -1;
      };
      int got = -1;
      // (a) the next tile of this XCD's run (block id = what the static map gives it)
      {
        const int k = draw(&queue[x], xcd_chunk);
        if (k >= 0) got = x + 8 * k;
      }
      // (b) a split unit, (c) a tile of another XCD's run
      if (got < 0) {
        const int u = draw(&queue[9], nunits);
        if (u >= 0) got = full_tiles + u;
      }
      for (int dx = 1; got < 0 && dx < 8; ++dx) {
        const int xx = (x + dx) & 7;
        const int k = draw(&queue[xx], xcd_chunk);
        if (k >= 0) got = xx + 8 * k;
      }
      *s_item = got;
    }
    __syncthreads();
    item_blk = *s_item;
    __syncthreads();  // smem is about to become the operand tiles
    if (item_blk < 0) break;  // nothing left
  }
  const bool whole = item_blk < full_tiles;
  const int unit = whole ? 0 : item_blk - full_tiles;  // index among the split-K units
  const int ksplit = whole ? 1 : ksplit_tail;
  const int chunk = unit % ksplit;
  int tile = whole ? item_blk : full_tiles + unit / ksplit;
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own
  // L2), so XCD x walks the contiguous run [x * xcd_chunk, (x + 1) * xcd_chunk) of the
  // patch-ordered tile list: the ~64 tiles it has in flight share 8 + 8 operand panels.
  if (!GROUPED && whole && xcd_chunk > 0) tile = (tile & 7) * xcd_chunk + (tile >> 3);
  if (!whole) stats.mode = 0;  // k_gemm_tail_stats covers the split tiles
  if (tilemap != nullptr) {
    const int2 t = tilemap[tile];
    ti = t.x;
    tj = t.y;
  } else {
    tile_coords<SYM>(tile, ntiles_m, ntiles_n, &ti, &tj);
  }
  const int row0 = ti * BM;
  const int col0 = tj * BN;

  // (opaque to the optimiser: otherwise every tid-derived address of the item body is hoisted
  //  out of the item loop and kept alive across the epilogue -- spills)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1;
  const int wc = wave & 1;
  const int li = lane & 15;
  const int lg = lane >> 4;
  // diagnostic (SC_GEMM_CLOCK=1, see launch_variant): shader cycles and wall ticks per tile
  long long clk0 = 0, wall0 = 0;
  const bool probe = whole && probe_out != nullptr;
  if (probe) {
    clk0 = clock64();
    wall0 = wall_clock64();
  }

  // --- global -> register staging: 4 chunks (16 B) of A and of B per thread.  Chunk q of a
  // thread is row (tid >> 3) + 32 q, k-chunk tid & 7; addresses are a per-tile base (the
  // tile's first row: buffer resource, SGPRs) plus a 32-bit byte offset per chunk (< 128 rows,
  // so it fits for any matrix size; rows clamped for ragged edge tiles).
  unsigned aoff[4], boff[4];
  int lds_off[4];
  const int kc0 = tid & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = (tid >> 3) + 32 * q;
    const int ra_ = row0 + r < M ? r : M - 1 - row0;
    const int rb_ = col0 + r < N ? r : N - 1 - col0;
    aoff[q] = (unsigned)(((size_t)ra_ * lda + 2 * kc0) * sizeof(double));
    boff[q] = (unsigned)(((size_t)rb_ * ldb + 2 * kc0) * sizeof(double));
    lds_off[q] = lds_chunk_off(r, kc0);
  }

  v4f64 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = (v4f64){0.0, 0.0, 0.0, 0.0};

  const int ktiles_all = (K + BK - 1) / BK;
  const int kper = (ktiles_all + ksplit - 1) / ksplit;
  const int kt_begin = chunk * kper;
  const int kt_end = min(ktiles_all, kt_begin + kper);
  double2 ra[4], rb[4];

  // Raw loads only: nothing here may consume the loaded registers, or hipcc waits
  // for the loads right away and the one-tile-ahead prefetch is lost.  The K-tail
  // guard is applied when the tile is written to LDS, one iteration later.
  // buffer_load with the K offset in an SGPR (soffset) and the row offset in one VGPR per
  // chunk: no vector ALU work per load inside the MFMA stream.
  const __amdgpu_buffer_rsrc_t arsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A + (size_t)row0 * lda), 0, -1,
                                        0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(B + (size_t)col0 * ldb), 0, -1,
                                        0x00020000);
  auto gload = [&](int kt) {
    const int koff = kt * BK * (int)sizeof(double);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4i32 va = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[q], koff, 0);
      const v4i32 vb = __builtin_amdgcn_raw_buffer_load_b128(brsrc, boff[q], koff, 0);
      ra[q] = __builtin_bit_cast(double2, va);
      rb[q] = __builtin_bit_cast(double2, vb);
    }
  };

  auto lds_store = [&](int buf, int kt) {
    const int k0 = kt * BK;
    if (k0 + BK > K) {  // wave-uniform: only the last K-tile can be ragged; zero in place
      const int k = k0 + 2 * kc0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (k >= K) { ra[q].x = 0.0; rb[q].x = 0.0; }
        if (k + 1 >= K) { ra[q].y = 0.0; rb[q].y = 0.0; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<double2*>(&As[buf][lds_off[q]]) = ra[q];
      *reinterpret_cast<double2*>(&Bs[buf][lds_off[q]]) = rb[q];
    }
  };

  // operand read offsets inside a tile: row part; position k ^ li added per sub-step
  const int arow = (wr * 64 + li) * BK;
  const int brow = (wc * 64 + li) * BK;

  // 1 for the workgroup that shares its CU with an earlier one (its LDS allocation does not
  // start at 0: HW_REG_LDS_ALLOC.LDS_BASE), see the K loop
  const int first_buf = (__builtin_amdgcn_s_getreg(6 | (31 << 11)) & 0xfff) != 0 ? 1 : 0;
  if (kt_begin < kt_end) {
    gload(kt_begin);
    lds_store(first_buf, kt_begin);
    if (kt_begin + 1 < kt_end) gload(kt_begin + 1);
  }
  __syncthreads();

  // Main loop.  Fragments are double-buffered in registers: the LDS reads of the next half
  // K-tile are issued before the 32 MFMAs of the current half, and the refill of the other
  // LDS buffer (ds_write of the tile fetched one iteration ago, then the global loads of the
  // tile after it) sits in the middle of the MFMA stream, so neither latency is exposed even
  // with one workgroup per CU (small n).  One barrier per K-tile: it orders the refill of
  // buffer cur^1 before the reads below it, and those reads (complete, __syncthreads waits for
  // lgkmcnt(0)) before the next iteration overwrites buffer cur.
  // k-slot lg of the two MFMA steps of a half carries k = 8p + 2lg and 8p + 2lg + 1: one
  // 16-byte LDS read per fragment feeds both steps (the order in which the K sum is taken is
  // free, as long as A and B agree on it).
  double2 fa0[4], fb0[4], fa1[4], fb1[4];
  auto load_frags = [&](double2 (&a)[4], double2 (&b)[4], const double* Ac, const double* Bc,
                        int p) {
    const int koff = (2 * (4 * p + lg)) ^ (li & 14);
#pragma unroll
    for (int m = 0; m < 4; ++m)
      a[m] = *reinterpret_cast<const double2*>(Ac + arow + m * 16 * BK + koff);
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
      b[nn] = *reinterpret_cast<const double2*>(Bc + brow + nn * 16 * BK + koff);
  };
  auto mfma16x = [&](const double2 (&a)[4], const double2 (&b)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
        acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[nn].x, acc[m][nn], 0, 0, 0);
  };
  auto mfma16y = [&](const double2 (&a)[4], const double2 (&b)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
        acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[nn].y, acc[m][nn], 0, 0, 0);
  };
  if (kt_begin < kt_end) load_frags(fa0, fb0, As[first_buf], Bs[first_buf], 0);
  // one K-tile; the LDS buffer index is a compile-time constant (the loop below is unrolled by
  // two), so every LDS address is a precomputed register + an immediate offset
  auto k_tile = [&](int kt, auto cur_c, auto prio_c) {
    constexpr int cur = decltype(cur_c)::value;
    constexpr int prio = decltype(prio_c)::value;
    load_frags(fa1, fb1, As[cur], Bs[cur], 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(prio);
    mfma16x(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < kt_end) {
      lds_store(cur ^ 1, kt + 1);
      if (kt + 2 < kt_end) gload(kt + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma16y(fa0, fb0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (kt + 1 < kt_end) load_frags(fa0, fb0, As[cur ^ 1], Bs[cur ^ 1], 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(prio);
    mfma16x(fa1, fb1);
    mfma16y(fa1, fb1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  long long wall_k0 = 0;
  if (probe) wall_k0 = wall_clock64();  // prologue done: K loop starts
  // The two workgroups of a CU take turns at the higher MFMA priority, one K-tile each: the
  // second one runs one peeled K-tile first (starting in LDS buffer 1), which shifts its
  // even / odd phase by one (measured: Diffuse -0.8 %).
  {
    using P0 = std::integral_constant<int, 2>;
    using P1 = std::integral_constant<int, 1>;
    int kt = kt_begin;
    if (first_buf && kt < kt_end) {
      k_tile(kt, std::integral_constant<int, 1>{}, P1{});
      ++kt;
    }
    for (; kt + 1 < kt_end; kt += 2) {
      k_tile(kt, std::integral_constant<int, 0>{}, P0{});
      k_tile(kt + 1, std::integral_constant<int, 1>{}, P1{});
    }
    if (kt < kt_end) k_tile(kt, std::integral_constant<int, 0>{}, P0{});
  }

  if (kEdgePrio & 1) __builtin_amdgcn_s_setprio(3);
  long long wall_k1 = 0;
  if (probe) wall_k1 = wall_clock64();
  if (probe && tid == 0) {
    probe_out[2 * item_blk] = (double)(clock64() - clk0);
    probe_out[2 * item_blk + 1] = (double)(wall_k1 - wall0);
    probe_out[2 * full_tiles + item_blk] = (double)wall0;
    probe_out[3 * full_tiles + item_blk] = (double)(wall_k0 - wall0);  // prologue ticks
  }
  // --- epilogue.  D layout of v_mfma_f64_16x16x4_f64: lane l, reg r holds
  //     D[row = (l >> 4) + 4 r][col = l & 15].
  if (ksplit > 1) {
    double* out = partial + (size_t)unit * (BM * BN);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((m * 4 + nn) * 4 + r) * 256 + tid] = acc[m][nn][r];
    continue;
  }
  const bool mirror = SYM && (ti != tj);
  constexpr bool nt_store = (kEdgePrio & 2) != 0;
  // wholly inside the matrix (wave-uniform): the epilogue without per-element guards
  const bool full = row0 + BM <= M && col0 + BN <= N;
  if (stats.mode != 0 || mirror) __syncthreads();  // operand tiles are dead: LDS is reused
  if (stats.mode != 0) {
    if (full) tile_row_stats_full<EPI, SYM>(acc, ti, tj, ntiles_n, tid, stats, smem);
    else tile_row_stats<EPI, SYM>(acc, ti, tj, ntiles_n, M, N, tid, stats, smem);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (probe && tid == 0) probe_out[4 * full_tiles + item_blk] = (double)(wall_clock64() - wall_k1);
  // The mirror tile is stored through a transposed copy in LDS (per wave, 16 rows of its 64 x 64
  // sub-tile at a time: T[c][r], pitch 20 doubles = two lanes per bank pair on the writes) so
  // that its stores are 128-byte row segments like the direct ones, not 8-byte scatters.
  constexpr int kStagePitch = 20;
  double* stage = smem + kEpiStage + wave * kEpiStageWave;
  if (full) {
    // one 64-bit address per 4-row group of the direct tile and per 8-row group of the mirror
    // tile, immediates for the rest; 16-byte stores for the mirror tile
    double* cdir = C + (size_t)(row0 + wr * 64 + lg) * ldc + (col0 + wc * 64 + li);
    const double* adir = nullptr;
    if (EPI == kEpiAdd)
      adir = stats.addend + (size_t)(row0 + wr * 64 + lg) * ldc + (col0 + wc * 64 + li);
    double* cmir = C + (size_t)(col0 + wc * 64 + (lane >> 3)) * ldc + (row0 + wr * 64 + 2 * (lane & 7));
    double* stage_w = stage + li * kStagePitch + lg;
    const double* stage_r = stage + (lane >> 3) * kStagePitch + 2 * (lane & 7);
    // (two straight-line instances: with `mirror` as a run-time test hipcc guards every LDS
    //  write with its own pair of scalar branches)
    auto store_full = [&](auto mirror_c) {
      constexpr bool MIRROR = decltype(mirror_c)::value;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t roff = (size_t)(m * 16 + 4 * r) * ldc;
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) {
            double v = acc[m][nn][r];
            if (EPI == kEpiAffinity) v = __builtin_fma(v, 0.5, 0.5);
            if (EPI == kEpiAdd) v += adir[roff + nn * 16];
            if (nt_store) __builtin_nontemporal_store(v, cdir + roff + nn * 16);
            else cdir[roff + nn * 16] = v;
            if (MIRROR) stage_w[nn * 16 * kStagePitch + 4 * r] = v;
          }
        }
        if (MIRROR) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const v2f64 v = *reinterpret_cast<const v2f64*>(stage_r + i * 8 * kStagePitch);
            v2f64* dst = reinterpret_cast<v2f64*>(cmir + (size_t)(i * 8) * ldc + m * 16);
            if (nt_store) __builtin_nontemporal_store(v, dst);
            else *dst = v;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    };
    if (mirror) store_full(std::true_type{});
    else store_full(std::false_type{});
  } else {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      const int col = col0 + wc * 64 + nn * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wr * 64 + m * 16 + lg + 4 * r;
        double v = acc[m][nn][r];
        // (v + 1) / 2: halving is exact, so one fma rounds identically
        if (EPI == kEpiAffinity) v = __builtin_fma(v, 0.5, 0.5);
        if (row < M && col < N) {
          if (EPI == kEpiAdd) v += stats.addend[(size_t)row * ldc + col];
          if (nt_store) __builtin_nontemporal_store(v, &C[(size_t)row * ldc + col]);
          else C[(size_t)row * ldc + col] = v;
        }
        if (mirror) stage[(nn * 16 + li) * kStagePitch + lg + 4 * r] = v;
      }
    }
    if (mirror) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = i * 8 + (lane >> 3);
        const int j = 2 * (lane & 7);
        const double2 v = *reinterpret_cast<const double2*>(stage + c * kStagePitch + j);
        const int mrow = col0 + wc * 64 + c;            // row of the mirror tile
        const int mcol = row0 + wr * 64 + m * 16 + j;   // its column (even)
        if (mrow < N) {
          double* dst = C + (size_t)mrow * ldc + mcol;
          if (mcol + 1 < M) {
            if (nt_store) {
              __builtin_nontemporal_store(v.x, dst);
              __builtin_nontemporal_store(v.y, dst + 1);
            } else {
              *reinterpret_cast<double2*>(dst) = v;
            }
          } else if (mcol < M) {
            dst[0] = v.x;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  }
  if (probe) {  // stores issued | stores drained (ticks since the end of the K loop)
    const long long t_issued = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      probe_out[5 * full_tiles + item_blk] = (double)(t_issued - wall_k1);
      probe_out[6 * full_tiles + item_blk] = (double)(wall_clock64() - wall_k1);
    }
  }
  }  // item loop
}

template <int EPI, bool SYM>
__global__ __launch_bounds__(256, 2) void k_gemm_nt(const double* __restrict__ A,
                                                 int lda,
                                                 const double* __restrict__ B,
                                                 int ldb, double* __restrict__ C,
                                                 int ldc, int M, int N, int K,
                                                 int ntiles_m, int ntiles_n,
                                                 int full_tiles, int ksplit_tail,
                                                 double* __restrict__ partial,
                                                 double* __restrict__ probe_out,
                                                 const int2* __restrict__ tilemap,
                                                 int xcd_chunk, GemmStats stats,
                                                 int* __restrict__ queue,
                                                 int nunits, int persist) {
  gemm_nt_body<EPI, SYM, false>(A, lda, B, ldb, C, ldc, M, N, K, ntiles_m, ntiles_n, full_tiles,
                                ksplit_tail, partial, probe_out, tilemap, xcd_chunk, stats,
                                queue, nunits, persist, nullptr);
}
// every workgroup: one whole tile of one member (no queue, no split, no probe)
template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_g(const GemmGroup grp, int stats_mode,
                                                     int xcd_chunk) {
  GemmStats stats{nullptr, nullptr, stats_mode, nullptr};
  gemm_nt_body<EPI, true, true>(nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, 0, 0, 0, 0x7fffffff, 1,
                                nullptr, nullptr, nullptr, xcd_chunk, stats, nullptr,
                                0, 0, &grp);
}

// rowmax / rowsum of every member from its per-tile partials (k_gemm_stats_reduce, grouped)
struct StatsReduceItem {
  const double* pmax;
  const double* psum;
  double* rowmax;
  double* rowsum;
  int n, nt;
};
__global__ __launch_bounds__(256) void k_gemm_stats_reduce_g(const GroupOf<StatsReduceItem> g,
                                                             int mode) {
  const StatsReduceItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x * kStatRows >= a.n) return;
  if (a.nt > kStatMaxTiles) {  // (very large members: one thread per row, straight from memory)
    const int i = blockIdx.x * kStatRows + threadIdx.x;
    if ((int)threadIdx.x >= kStatRows || i >= a.n) return;
    double mx = -INFINITY, sm = 0.0;
    for (int t = 0; t < a.nt; ++t) {
      mx = fmax(mx, a.pmax[(size_t)i * a.nt + t]);
      if (mode == 1) sm += a.psum[(size_t)i * a.nt + t];
    }
    if (mode == 2) mx = fmax(mx, 0.0);
    a.rowmax[i] = mx;
    if (mode == 1) a.rowsum[i] = sm;
    return;
  }
  gemm_stats_reduce_lds_body(a.pmax, a.psum, a.n, a.nt, mode, a.rowmax, a.rowsum);
}

// Sums the ksplit partial tiles of k_gemm_nt (fixed order: deterministic), applies
// the epilogue and stores the tile (+ mirror).  Same fragment -> (row, col) map.
template <int EPI, bool SYM>
__global__ __launch_bounds__(256) void k_gemm_reduce(const double* __restrict__ partial,
                                                     double* __restrict__ C, int ldc,
                                                     int M, int N, int ntiles_m,
                                                     int ntiles_n, int tile_offset,
                                                     int ksplit,
                                                     const int2* __restrict__ tilemap,
                                                     const double* __restrict__ addend) {
  int ti, tj;
  if (tilemap != nullptr) {
    const int2 t = tilemap[tile_offset + blockIdx.x];
    ti = t.x;
    tj = t.y;
  } else {
    tile_coords<SYM>(tile_offset + blockIdx.x, ntiles_m, ntiles_n, &ti, &tj);
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const bool mirror = SYM && (ti != tj);
  const double* base = partial + (size_t)blockIdx.x * ksplit * (BM * BN);
  // blockIdx.y picks one of the 16 (m, nn) fragment pairs: 16x more workgroups
  {
    const int m = blockIdx.y >> 2;
    {
      const int nn = blockIdx.y & 3;
      const int col = tj * BN + wc * 64 + nn * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * BM + wr * 64 + m * 16 + lg + 4 * r;
        const int slot = ((m * 4 + nn) * 4 + r) * 256 + tid;
        double v = 0.0;
        for (int c = 0; c < ksplit; ++c) v += base[(size_t)c * (BM * BN) + slot];
        if (EPI == kEpiAffinity) v = __builtin_fma(v, 0.5, 0.5);
        if (row < M && col < N) {
          if (EPI == kEpiAdd) v += addend[(size_t)row * ldc + col];
          C[(size_t)row * ldc + col] = v;
          if (mirror) C[(size_t)col * ldc + row] = v;
        }
      }
    }
  }
}

// Row statistics for the split-K tail tiles, read back from C once k_gemm_reduce has
// written them (they are L2-hot).  blockIdx.y = 0: rows of tile (ti, tj) -> slot tj;
// blockIdx.y = 1 (SYM): rows of the mirror tile (tj, ti) -> slot ti.
template <bool SYM>
__global__ __launch_bounds__(256) void k_gemm_tail_stats(const double* __restrict__ C, int ldc,
                                                         int M, int N, int ntiles_m,
                                                         int ntiles_n, int tile_base,
                                                         const int2* __restrict__ tilemap,
                                                         GemmStats st) {
  int ti, tj;
  if (tilemap != nullptr) {
    const int2 t = tilemap[tile_base + blockIdx.x];
    ti = t.x;
    tj = t.y;
  } else {
    tile_coords<SYM>(tile_base + blockIdx.x, ntiles_m, ntiles_n, &ti, &tj);
  }
  if (blockIdx.y == 1) {
    if (!SYM || ti == tj) return;
    const int t = ti;
    ti = tj;
    tj = t;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // blockIdx.z: 8 groups of 16 rows, so the few tail tiles still spread over the chip
  for (int lr = blockIdx.z * 16 + wave; lr < blockIdx.z * 16 + 16; lr += 4) {
    const int row = ti * BM + lr;
    if (row >= M) break;
    double mx = -INFINITY, sm = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = tj * BN + lane + 64 * h;
      if (col < N) {
        const double x = C[(size_t)row * ldc + col];
        if (!(st.mode == 2 && row == col)) mx = fmax(mx, x);
        sm += x;
      }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      mx = fmax(mx, __shfl_xor(mx, o));
      sm += __shfl_xor(sm, o);
    }
    if (lane == 0) {
      st.pmax[(size_t)row * ntiles_n + tj] = mx;
      if (st.mode == 1) st.psum[(size_t)row * ntiles_n + tj] = sm;
    }
  }
}

__global__ void k_gemm_queue_init(int* queue) { queue[threadIdx.x] = 0; }

// co-resident k_gemm_nt workgroups on the current device (occupancy x CUs)
int gemm_resident_slots() {
  static int slots_dev[16] = {0};
  int dev = 0;
  hipGetDevice(&dev);
  dev = dev < 16 ? dev : 0;
  if (slots_dev[dev] == 0) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &per_cu, reinterpret_cast<const void*>(k_gemm_nt<kEpiNone, true>), 256, 0);
    if (per_cu < 1) per_cu = 1;
    slots_dev[dev] = per_cu * prop.multiProcessorCount;
  }
  return slots_dev[dev];
}
size_t gemm_splitk_workspace_bytes() {
  return (size_t)gemm_resident_slots() * BM * BN * sizeof(double);
}

template <int EPI, bool SYM>
static void launch_variant(hipStream_t s, const double* A, int lda, const double* B,
                           int ldb, double* C, int ldc, int M, int N, int K,
                           double* splitk_ws, const int2* tilemap, const GemmRowStats* rs,
                           const double* addend) {
  GemmStats stats{nullptr, nullptr, 0, addend};
  if (rs != nullptr && rs->mode != 0) {
    stats.pmax = rs->partial_max;
    stats.psum = rs->partial_sum;
    stats.mode = rs->mode;
  }
  const int tm = (M + BM - 1) / BM;
  const int tn = (N + BN - 1) / BN;
  const int tiles = SYM ? tm * (tm + 1) / 2 : tm * tn;
  const int g_slots = gemm_resident_slots();
  double* g_partial = splitk_ws;  // per-handle scratch: handles may run concurrently
  const int ktiles = (K + BK - 1) / BK;
  // full waves of workgroups run whole tiles; the ragged remainder is split over K
  int full = (tiles / g_slots) * g_slots;
  int rem = tiles - full;
  int ksplit = 1;
  if (rem > 0) {
    if (splitk_ws != nullptr) {
      ksplit = g_slots / rem;
      ksplit = std::min(ksplit, std::max(1, ktiles / 8));  // >= 8 k-tiles per chunk
      // a short product that does not fill the chip anyway (an utterance of a few thousand
      // rows against d = 256 features): the K loop of a whole tile is a third of the tile's
      // epilogue, splitting it only adds the partial stores and two more launches
      if (K <= 512 && full == 0) ksplit = 1;
    }
    if (ksplit < 2) {  // no workspace, or not worth splitting: whole tiles only
      full = tiles;
      rem = 0;
      ksplit = 1;
    }
  }
  {
    const int xcd_chunk = (tilemap != nullptr && full % 8 == 0 && full >= 512) ? full / 8 : 0;
    // SC_GEMM_CLOCK=1 (diagnostic): every launch also records, per whole tile, the
    // shader-clock cycles (s_memtime) and the constant-rate wall ticks (s_memrealtime)
    // between kernel entry and the end of the K loop, synchronises and prints the effective
    // shader clock -- the GEMM is power-managed, see DESIGN_HISTORY.md section 3.3.  Any other value is
    // a file that also receives "workgroup cycles ticks start_tick ..." per tile
    // (tools/gemm_tile_timeline.py reads it).  Off: one pointer compare per workgroup.
    static double* dbg = nullptr;
    static const char* probe_env = getenv("SC_GEMM_CLOCK");
    if (probe_env != nullptr && dbg == nullptr) (void)hipMalloc(&dbg, sizeof(double) * 7 * 8192);
    double* probe = (full > 0 && full <= 8192) ? dbg : nullptr;
    // Long K (the Diffuse product): persistent workgroups drawing items from a queue ([0..7]
    // next tile per XCD, [9] next split unit) -- no teardown / dispatch between tiles.  With
    // K = 256 a tile is 16 K-tiles and the draw costs more than the dispatch it saves
    // (affinity GEMM 0.395 -> 0.417 ms): one workgroup per item by block index there.
    static int* queue_buf[16] = {nullptr};
    int* queue = nullptr;
    if (K >= 1024 && xcd_chunk > 0 && rem > 0 && SYM) {
      int dev = 0;
      hipGetDevice(&dev);
      dev &= 15;
      if (queue_buf[dev] == nullptr) (void)hipMalloc(&queue_buf[dev], 16 * sizeof(int));
      queue = queue_buf[dev];
      hipLaunchKernelGGL(k_gemm_queue_init, dim3(1), dim3(16), 0, s, queue);
    }
    const int persist = queue != nullptr ? 1 : 0;
    const int grid = persist ? std::min(g_slots, full + rem * ksplit) : full + rem * ksplit;
    hipLaunchKernelGGL((k_gemm_nt<EPI, SYM>), dim3(grid), dim3(256), 0, s, A, lda,
                       B, ldb, C, ldc, M, N, K, tm, tn, full, ksplit, g_partial, probe, tilemap,
                       xcd_chunk, stats, queue, rem * ksplit, persist);
    if (probe != nullptr) {
      (void)hipStreamSynchronize(s);
      std::vector<double> h(7 * full);
      (void)hipMemcpy(h.data(), dbg, sizeof(double) * 7 * full, hipMemcpyDeviceToHost);
      if (strcmp(probe_env, "1") != 0) {
        const char* path = probe_env;
        // workgroup, cycles, ticks entry..K-loop end, start tick, prologue ticks, then ticks
        // since the end of the K loop: after the row statistics, stores issued, stores drained
        if (FILE* f = fopen(path, "w")) {
          for (int b = 0; b < full; ++b)
            fprintf(f, "%d %.0f %.0f %.0f %.0f %.0f %.0f %.0f\n", b, h[2 * b], h[2 * b + 1],
                    h[2 * full + b], h[3 * full + b], h[4 * full + b], h[5 * full + b],
                    h[6 * full + b]);
          fclose(f);
        }
      }
      double c = 0, w = 0;
      for (int i = 0; i < full; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
      int rate = 0;
      (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
      fprintf(stderr, "gemm clock probe (epilogue %d%s): %d tiles, K %d, shader cycles/tile %.0f "
              "(MFMA-bound minimum %.0f), effective shader clock %.1f MHz, tile time %.1f us\n",
              EPI, SYM ? ", symmetric" : "", full, K, c / full,
              (double)((K + BK - 1) / BK) * 64 * 64 * 2, c / w * rate / 1e3,
              w / full / rate * 1e3);
    }
  }
  if (rem > 0) {
    hipLaunchKernelGGL((k_gemm_reduce<EPI, SYM>), dim3(rem, 16), dim3(256), 0, s, g_partial,
                       C, ldc, M, N, tm, tn, full, ksplit, tilemap, addend);
    if (stats.mode != 0)
      hipLaunchKernelGGL((k_gemm_tail_stats<SYM>), dim3(rem, SYM ? 2 : 1, 8), dim3(256), 0, s, C,
                         ldc, M, N, tm, tn, full, tilemap, stats);
  }
  if (stats.mode != 0) {
    if (tn <= kStatMaxTiles)  // (n <= 15360: the staged rows fit the default LDS limit)
      hipLaunchKernelGGL(k_gemm_stats_reduce_lds, dim3((M + kStatRows - 1) / kStatRows), dim3(256),
                         2 * kStatRows * (size_t)(tn + 1) * sizeof(double), s, stats.pmax,
                         stats.psum, M, tn, stats.mode, rs->rowmax, rs->rowsum);
    else
      hipLaunchKernelGGL(k_gemm_stats_reduce, dim3((M + 255) / 256), dim3(256), 0, s, stats.pmax,
                         stats.psum, M, tn, stats.mode, rs->rowmax, rs->rowsum);
  }
}

// Upper-triangle tiles (ti <= tj) of an nt x nt tile grid in patch order: 8 x 8-tile
// patches row by row, tiles row by row inside a patch.  Returns nt (nt + 1) / 2 pairs.
void gemm_build_sym_tilemap(int nt, std::vector<int2>* out) {
  out->clear();
  const int np = (nt + 7) / 8;
  for (int pi = 0; pi < np; ++pi)
    for (int pj = pi; pj < np; ++pj)
      for (int ti = pi * 8; ti < std::min(nt, pi * 8 + 8); ++ti)
        for (int tj = std::max(ti, pj * 8); tj < std::min(nt, pj * 8 + 8); ++tj)
          out->push_back(make_int2(ti, tj));
}
int gemm_tile_dim(int n) { return (n + BM - 1) / BM; }

void launch_gemm_nt(hipStream_t s, const double* A, int lda, const double* B,
                    int ldb, double* C, int ldc, int M, int N, int K,
                    int epilogue, bool symmetric, double* splitk_ws,
                    const int2* tilemap, const GemmRowStats* rs, const double* addend) {
  if (M <= 0 || N <= 0) return;
#define SC_GEMM_CASE(E, S)                                                                \
  launch_variant<E, S>(s, A, lda, B, ldb, C, ldc, M, N, K, splitk_ws, S ? tilemap : nullptr, \
                       rs, addend)
  if (symmetric) {
    if (epilogue == kEpiAffinity) SC_GEMM_CASE(kEpiAffinity, true);
    else if (epilogue == kEpiAdd) SC_GEMM_CASE(kEpiAdd, true);
    else SC_GEMM_CASE(kEpiNone, true);
  } else {
    if (epilogue == kEpiAffinity) SC_GEMM_CASE(kEpiAffinity, false);
    else if (epilogue == kEpiAdd) SC_GEMM_CASE(kEpiAdd, false);
    else SC_GEMM_CASE(kEpiNone, false);
  }
#undef SC_GEMM_CASE
}

// C_z = A_z A_z^T (+ the affinity epilogue) of every member in ONE launch, row statistics from
// the tile epilogues (`stats_mode` 1: row max and sum, 2: CropDiagonal's value) reduced by one
// more grouped launch.  items[z].n = 0: idle member.
void launch_gemm_nt_group(hipStream_t s, const GemmGroupItem* items, int count, int epilogue,
                          int stats_mode) {
  GemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  GroupOf<StatsReduceItem> red;
  memset(&red, 0, sizeof(red));
  int total = 0, nmax = 0;
  for (int z = 0; z < kGroupMax; ++z) {
    grp.first[z] = total;
    if (z >= count || items[z].n <= 0) continue;
    const GemmGroupItem& it = items[z];
    const int nt = (it.n + BM - 1) / BM;
    grp.m[z] = GemmMember{it.A, it.C, it.partial_max, it.partial_sum, it.tilemap,
                          it.lda, it.ldc, it.n, it.K, nt};
    red.s[z] = StatsReduceItem{it.partial_max, it.partial_sum, it.rowmax, it.rowsum, it.n, nt};
    total += nt * (nt + 1) / 2;
    nmax = std::max(nmax, it.n);
  }
  grp.first[kGroupMax] = total;
  if (total == 0) return;
  // one workgroup per tile (a persistent stride loop over the tile list measured 2 % slower on
  // config 5: no balancing between slots, spilled loop-carried state; not kept)
  const int xcd_chunk = (total + 7) / 8;
  const int grid = 8 * xcd_chunk;
  if (epilogue == kEpiAffinity)
    hipLaunchKernelGGL((k_gemm_nt_g<kEpiAffinity>), dim3(grid), dim3(256), 0, s, grp, stats_mode,
                       xcd_chunk);
  else
    hipLaunchKernelGGL((k_gemm_nt_g<kEpiNone>), dim3(grid), dim3(256), 0, s, grp, stats_mode,
                       xcd_chunk);
  if (stats_mode != 0)
    hipLaunchKernelGGL(k_gemm_stats_reduce_g, dim3((nmax + kStatRows - 1) / kStatRows, count),
                       dim3(256),
                       2 * kStatRows * (size_t)(std::min((nmax + BM - 1) / BM, kStatMaxTiles) + 1) *
                           sizeof(double),
                       s, red, stats_mode);
}

}  // namespace sc
