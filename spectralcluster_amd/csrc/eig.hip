// Symmetric top-k eigensolver kernels (reference utils.py:44-71 calls LAPACK
// dgeev on a matrix that is diagonally similar to a symmetric one).
//
// Operator:  Op = diag(p) + diag(c) S diag(c),  S = refined (symmetric) matrix.
// Method:    block Lanczos, 8 vectors per block (half a v_mfma_f64_16x16x4_f64 tile column),
//            full re-orthogonalisation, explicit Rayleigh-Ritz T = Q^T Op Q, thick restart.
//   k_block_matvec   the only O(n^2) kernel: one HBM pass over S per block
//   k_lz_rows        the orthonormalisation as a chain of short launches (4 per block) whose
//                    prologues add the previous link's partial sums: no host sync per block
//   k_proj_partial / k_reduce_H / k_update_block / k_reduce_chol / k_apply_rinv
//                    the host-driven form of the same chain (one step per launch, flags read
//                    after every block): the repair path for rank-deficient blocks
//   k_jacobi         one-workgroup cyclic Jacobi, matrix in LDS: the dense solver for n <= 128
//                    and the Rayleigh-Ritz problems above 64 (smaller ones are solved on the
//                    host, eig_driver.hip)
// Everything except the matvec is tall-skinny (n x <= 136), L2-resident and latency-bound.
#include <algorithm>
#include <cstring>
#include <mutex>

#include "sc_internal.h"

namespace sc {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int B = kEigBlock;     // vectors per block (8)
constexpr int RB = 256 / B;       // rows a 256-thread workgroup covers in row-parallel kernels
static_assert(B == 8 || B == 16, "block width must divide the 16-wide MFMA tile");

// ---------------------------------------------------------------- random block
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ double hash_uniform(uint64_t seed, uint64_t idx) {
  const uint64_t h = splitmix64(seed ^ splitmix64(idx));
  return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;  // [-1, 1)
}

__global__ void k_random_block(double* W, int n, uint64_t seed) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n * B) W[e] = hash_uniform(seed, (uint64_t)e);
}

__global__ void k_refill_deficient(double* W, int n, const int* flags,
                                   uint64_t seed) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * B) return;
  if ((flags[0] >> (e & (B - 1))) & 1) W[e] = hash_uniform(seed, (uint64_t)e);
}

// ---------------------------------------------------------------- block matvec
// W[r, :] = p[r] * V[r, :] + c[r] * sum_k S[r, k] * Vs[k, :]   (Vs = c .* V)
// One workgroup = 16 rows of S; its 8 waves split K in interleaved 32-wide chunks.
// Lane (i = l & 15, g = l >> 4) loads S[r0 + i][kb + 8 g .. + 7] (64 B contiguous,
// next chunk prefetched one iteration ahead) and feeds it to 8 MFMAs whose k-slot g
// carries k = kb + 8 g + t; two accumulators break the MFMA dependency chain.
// HBM-bound: one pass over S (n^2 * 8 bytes) per 16 vectors.
constexpr int kMvWaves = 8;
__device__ __forceinline__ void block_matvec_body(
    const double* __restrict__ S, int ld, int n, const double* __restrict__ cvec,
    const double* __restrict__ pvec, const double* __restrict__ V, int ldv,
    const double* __restrict__ Vs, double* __restrict__ W) {
  __shared__ double red[kMvWaves][64][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int r0 = blockIdx.x * 16;
  int ri = r0 + li;
  ri = ri < n ? ri : n - 1;
  // k-slot map inside a 32-column chunk: lane group lg, step q holds columns 8 q + 2 lg, +1 (the
  // order of the K sum is free as long as A and B agree), so that the four lane groups of one
  // load instruction read 64 contiguous bytes of their row -- whole sectors, not 16-byte pieces
  const double* srow = S + (size_t)ri * ld + 2 * lg;
  v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  const int nchunks = (n + 31) / 32;
  double2 cur[4], nxt[4];
  auto load = [&](double2* dst, int ch) {
    const int kb = ch * 32 + 2 * lg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // rows are padded to ld (multiple of 16): never read past the row's storage
      if (kb + 8 * q + 1 < ld)
        dst[q] = *reinterpret_cast<const double2*>(srow + ch * 32 + 8 * q);
      else
        dst[q] = make_double2(0.0, 0.0);
    }
  };
  int ch = wave;
  if (ch < nchunks) load(cur, ch);
  for (; ch < nchunks; ch += kMvWaves) {
    if (ch + kMvWaves < nchunks) load(nxt, ch + kMvWaves);
    const int kb = ch * 32 + 2 * lg;
    double b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // MFMA tile is 16 wide; columns >= B carry zeros
      const int k = kb + 8 * (t >> 1) + (t & 1);
      b[t] = (k < n && li < B) ? Vs[(size_t)k * B + li] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double a0 = (kb + 8 * q < n) ? cur[q].x : 0.0;
      const double a1 = (kb + 8 * q + 1 < n) ? cur[q].y : 0.0;
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[2 * q], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[2 * q + 1], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc0[r] + acc1[r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < kMvWaves; ++w) sum += red[w][lane][r];
      const int row = r0 + lg + 4 * r;  // D[row = (l >> 4) + 4 r][col = l & 15]
      // (cvec == nullptr: c = 1, pvec == nullptr: p = 0 -- the plain product A Vs, the first
      //  half of the matrix-free Diffuse operator)
      if (row < n && li < B)
        W[(size_t)row * B + li] =
            __builtin_fma(cvec ? cvec[row] : 1.0, sum,
                          pvec ? pvec[row] * V[(size_t)row * ldv + li] : 0.0);
    }
  }
}
__global__ __launch_bounds__(64 * kMvWaves) void k_block_matvec(
    const double* __restrict__ S, int ld, int n, const double* __restrict__ cvec,
    const double* __restrict__ pvec, const double* __restrict__ V, int ldv,
    const double* __restrict__ Vs, double* __restrict__ W) {
  block_matvec_body(S, ld, n, cvec, pvec, V, ldv, Vs, W);
}
// Grouped form (batch_group.hip): blockIdx.y picks one of up to kGroupMax independent
// problems whose descriptors travel in the kernel arguments; a problem with n = 0 is idle.
__global__ __launch_bounds__(64 * kMvWaves) void k_block_matvec_g(const GroupOf<MatvecItem> g) {
  const MatvecItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x * 16 >= a.n) return;
  block_matvec_body(a.S, a.ld, a.n, a.cvec, a.pvec, a.V, a.ldv, a.Vs, a.W);
}

// ---------------------------------------------------------------- symmetric block matvec
// The same product from the UPPER TRIANGLE of S only: half the HBM bytes per pass.
// One workgroup = one 128 x 128 tile (I, J), J >= I, of S.  It contributes
//   direct :  U[I rows]  += S_IJ   * Vs[J rows]
//   mirror :  U[J rows]  += S_IJ^T * Vs[I rows]        (J > I only)
// Each wave owns 32 rows of the tile.  It loads them in the direct product's MFMA layout
// (lane (li, lg): row li, 16-byte column chunks -- the same sector-aligned k-slot map as
// k_block_matvec), feeds the direct MFMAs, then turns ITS OWN 32 x 32 sub-block around
// through a wave-private LDS patch (row-major write, column-wise read: lane (i, g) reads
// S[4t + g][i], the A operand of the transposed product) -- no workgroup barrier in the loop.
// The two 128 x 8 results of a tile go to per-tile partial slabs; k_matvec_sym_reduce adds,
// for every block row R, the direct slabs (R, R..) and the mirror slabs (0..R-1, R) in that
// fixed order and applies  W = p .* V + c .* U.  Extra traffic: 2 x 8 KB per 128 KB tile,
// written and read once (+25 % of the halved matrix traffic).
constexpr int kSymTile = 128;
constexpr int kSymPitch = 34;  // doubles per row of a wave's 32 x 32 LDS patch
__device__ __forceinline__ int sym_item_id(int I, int J, int nt) {
  return I * nt - I * (I - 1) / 2 + (J - I);
}
__device__ __forceinline__ void block_matvec_sym_body(
    const double* __restrict__ S, int ld, int n, const double* __restrict__ Vs,
    double* __restrict__ pdirect, double* __restrict__ pmirror, int nt) {
  __shared__ __attribute__((aligned(16))) double smem[4 * 32 * kSymPitch];
  // block -> (I, J): row-major over the upper triangle
  int I = 0, rem = blockIdx.x, rowlen = nt;
  while (rem >= rowlen) {
    rem -= rowlen;
    --rowlen;
    ++I;
  }
  const int J = I + rem;
  const bool diag = I == J;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int r0 = I * kSymTile + 32 * wave;   // this wave's 32 rows
  const int c0 = J * kSymTile;
  double* patch = smem + wave * (32 * kSymPitch);
  // B operand of the mirror product: Vs rows of this wave, k-step t carries rows 4t + g
  double bm[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int row = r0 + 4 * t + lg;
    bm[t] = (row < n && li < B) ? Vs[(size_t)row * B + li] : 0.0;
  }
  v4f64 accd[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  v4f64 accm[4][2];
#pragma unroll
  for (int cs = 0; cs < 4; ++cs)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) accm[cs][ct] = (v4f64){0.0, 0.0, 0.0, 0.0};
  // (no register double-buffering of the strip: at ~120 VGPRs four workgroups share a CU and
  //  cover each other's load latency -- measured faster than 2 workgroups with prefetch)
  double2 a[1][2][4];  // [buffer][row tile][q]
  auto load = [&](int buf, int cs) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      int row = r0 + 16 * rt + li;
      row = row < n ? row : n - 1;
      const double* src = S + (size_t)row * ld + c0 + 32 * cs + 2 * lg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // rows are padded to ld (multiple of 16): never read past the row's storage
        if (c0 + 32 * cs + 8 * q + 2 * lg + 1 < ld)
          a[buf][rt][q] = *reinterpret_cast<const double2*>(src + 8 * q);
        else
          a[buf][rt][q] = make_double2(0.0, 0.0);
      }
    }
  };
#pragma unroll
  for (int cs = 0; cs < 4; ++cs) {
    constexpr int cur = 0;
    load(cur, cs);
    const int kb = c0 + 32 * cs + 2 * lg;
    // zero what lies outside the matrix (rows >= n were clamped, columns >= n are padding)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const bool rok = r0 + 16 * rt + li < n;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!rok || kb + 8 * q >= n) a[cur][rt][q].x = 0.0;
        if (!rok || kb + 8 * q + 1 >= n) a[cur][rt][q].y = 0.0;
      }
    }
    // ---- direct product
    double b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = kb + 8 * (t >> 1) + (t & 1);
      b[t] = (k < n && li < B) ? Vs[(size_t)k * B + li] : 0.0;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        accd[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][rt][q].x, b[2 * q], accd[rt], 0, 0, 0);
        accd[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][rt][q].y, b[2 * q + 1], accd[rt], 0, 0, 0);
      }
    if (!diag) {
      // ---- mirror product: this wave's 32 x 32 sub-block, turned around through LDS
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<double2*>(patch + (16 * rt + li) * kSymPitch + 8 * q + 2 * lg) =
              a[cur][rt][q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const double at = patch[(4 * t + lg) * kSymPitch + 16 * ct + li];
          accm[cs][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(at, bm[t], accm[cs][ct], 0, 0, 0);
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  // ---- slabs.  D layout: lane l, reg r holds D[row = (l >> 4) + 4 r][col = l & 15]
  const size_t item = blockIdx.x;
  if (li < B) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pdirect[(item * kSymTile + 32 * wave + 16 * rt + lg + 4 * r) * B + li] = accd[rt][r];
  }
  if (diag) return;
  // mirror: the four waves' contributions to the tile's 128 columns, added wave 0..3
  __syncthreads();  // the patches are dead: LDS becomes the reduction buffer [4][128][B]
  double* red = smem;
  if (li < B) {
#pragma unroll
    for (int cs = 0; cs < 4; ++cs)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          red[(wave * kSymTile + 32 * cs + 16 * ct + lg + 4 * r) * B + li] = accm[cs][ct][r];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kSymTile * B; e += 256)
    pmirror[item * (kSymTile * B) + e] =
        ((red[e] + red[kSymTile * B + e]) + red[2 * kSymTile * B + e]) + red[3 * kSymTile * B + e];
}
__global__ __launch_bounds__(256) void k_block_matvec_sym(
    const double* __restrict__ S, int ld, int n, const double* __restrict__ Vs,
    double* __restrict__ pdirect, double* __restrict__ pmirror, int nt) {
  block_matvec_sym_body(S, ld, n, Vs, pdirect, pmirror, nt);
}
__global__ __launch_bounds__(256) void k_block_matvec_sym_g(const GroupOf<MatvecItem> g) {
  const MatvecItem& a = g.s[blockIdx.y];
  const int nt = (a.n + kSymTile - 1) / kSymTile;
  if ((int)blockIdx.x >= nt * (nt + 1) / 2) return;  // (n = 0: idle member)
  block_matvec_sym_body(a.S, a.ld, a.n, a.Vs, a.slabs, a.slabs + (size_t)(nt * (nt + 1) / 2) * kSymTile * B, nt);
}

// W[r, :] = p[r] V[r, :] + c[r] * (sum of the slabs of block row R = r / 128, fixed order)
__device__ __forceinline__ void matvec_sym_reduce_body(
    const double* __restrict__ pdirect, const double* __restrict__ pmirror, int nt, int n,
    const double* __restrict__ cvec, const double* __restrict__ pvec,
    const double* __restrict__ V, int ldv, double* __restrict__ W) {
  const int R = blockIdx.x / 4;                       // 4 workgroups per block row
  const int e = (blockIdx.x & 3) * 256 + threadIdx.x;  // entry of the 128 x B slab
  const int rl = e / B, v = e % B;
  const int row = R * kSymTile + rl;
  double acc = 0.0;
  // direct slabs (R, J), J = R .. nt-1, then mirror slabs (I, R), I = 0 .. R-1; loads in
  // independent batches of 8
  const int total = nt;  // (nt - R) + R
  for (int s0 = 0; s0 < total; s0 += 8) {
    double x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int sidx = s0 + u;
      x[u] = 0.0;
      if (sidx < nt - R) {
        x[u] = pdirect[(size_t)sym_item_id(R, R + sidx, nt) * (kSymTile * B) + e];
      } else if (sidx < total) {
        x[u] = pmirror[(size_t)sym_item_id(sidx - (nt - R), R, nt) * (kSymTile * B) + e];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += x[u];
  }
  if (row < n) W[(size_t)row * B + v] =
      __builtin_fma(cvec ? cvec[row] : 1.0, acc, pvec ? pvec[row] * V[(size_t)row * ldv + v] : 0.0);
}
__global__ __launch_bounds__(256) void k_matvec_sym_reduce(
    const double* __restrict__ pdirect, const double* __restrict__ pmirror, int nt, int n,
    const double* __restrict__ cvec, const double* __restrict__ pvec,
    const double* __restrict__ V, int ldv, double* __restrict__ W) {
  matvec_sym_reduce_body(pdirect, pmirror, nt, n, cvec, pvec, V, ldv, W);
}
__global__ __launch_bounds__(256) void k_matvec_sym_reduce_g(const GroupOf<MatvecItem> g) {
  const MatvecItem& a = g.s[blockIdx.y];
  const int nt = (a.n + kSymTile - 1) / kSymTile;
  if ((int)blockIdx.x >= nt * 4) return;
  matvec_sym_reduce_body(a.slabs, a.slabs + (size_t)(nt * (nt + 1) / 2) * kSymTile * B, nt, a.n,
                         a.cvec, a.pvec, a.V, a.ldv, a.W);
}

// ---------------------------------------------------------------- projections
// partial[blk][i * B + j] = sum over the block's rows of Q[r][i] * W[r][j]
constexpr int kProjGroups = 256 / B;                          // i-groups per workgroup
constexpr int kProjAcc = (kLdq + kProjGroups - 1) / kProjGroups;
__global__ __launch_bounds__(256) void k_proj_partial(
    const double* __restrict__ Q, int ldq, int m, const double* __restrict__ W,
    int n, double* __restrict__ partial) {
  __shared__ double Ql[8][kLdq];
  __shared__ double Wl[8][B];
  const int tid = threadIdx.x;
  const int jj = tid % B, ig = tid / B;
  const int rows_per = (n + gridDim.x - 1) / gridDim.x;
  const int rbeg = blockIdx.x * rows_per;
  const int rend = min(n, rbeg + rows_per);
  double acc[kProjAcc];
#pragma unroll
  for (int q = 0; q < kProjAcc; ++q) acc[q] = 0.0;
  for (int r = rbeg; r < rend; r += 8) {
    __syncthreads();
    for (int e = tid; e < 8 * m; e += 256) {
      const int rr = e / m, i = e - rr * m;
      Ql[rr][i] = (r + rr < rend) ? Q[(size_t)(r + rr) * ldq + i] : 0.0;
    }
    if (tid < 8 * B) {
      const int rr = tid / B;
      Wl[rr][jj] = (r + rr < rend) ? W[(size_t)(r + rr) * B + jj] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const double w = Wl[rr][jj];
#pragma unroll
      for (int q = 0; q < kProjAcc; ++q)
        if (ig + kProjGroups * q < m)
          acc[q] = __builtin_fma(Ql[rr][ig + kProjGroups * q], w, acc[q]);
    }
  }
  double* out = partial + (size_t)blockIdx.x * (kLdq * B);
#pragma unroll
  for (int q = 0; q < kProjAcc; ++q)
    if (ig + kProjGroups * q < m) out[(ig + kProjGroups * q) * B + jj] = acc[q];
}

// H = sum of partials (m x 16), stored to Hbuf; optionally accumulated into
// T[0:m, col0:col0+16] and mirrored (upper triangle of the diagonal block only);
// hsq[j] += sum_i H_ij^2 (atomic: only feeds the rank-deficiency threshold; the
// caller zeroes hsq before the first pass).  One entry per thread, m*16/256 blocks.
__global__ __launch_bounds__(256) void k_reduce_H(
    const double* __restrict__ partial, int nparts, int m, double* __restrict__ Hbuf,
    double* __restrict__ T, int ldt, int col0, int accumulate,
    double* __restrict__ hsq) {
  // 8 lanes per entry: each sums every 8th partial, then a fixed-order butterfly
  __shared__ double sq[32];
  const int tid = threadIdx.x;
  const int sub = tid & 7;
  const int e = blockIdx.x * 32 + (tid >> 3);
  double h = 0.0;
  if (e < m * B)
    for (int g = sub; g < nparts; g += 8) h += partial[(size_t)g * (kLdq * B) + e];
  h += __shfl_xor(h, 1);
  h += __shfl_xor(h, 2);
  h += __shfl_xor(h, 4);
  if (sub == 0) {
    if (e < m * B) {
      Hbuf[e] = h;
      if (T != nullptr) {
        const int i = e / B, j = col0 + (e % B);
        if (i <= j) {
          const double v = accumulate ? T[(size_t)i * ldt + j] + h : h;
          T[(size_t)i * ldt + j] = v;
          T[(size_t)j * ldt + i] = v;
        }
      }
    } else {
      h = 0.0;
    }
    sq[tid >> 3] = h * h;
  }
  __syncthreads();
  if (tid < B) {  // entry e = 32 * block + q has column e % B == q % B (32 % B == 0)
    double s = 0.0;
    for (int q = tid; q < 32; q += B) s += sq[q];
    atomicAdd(&hsq[tid], s);
  }
}

// W[r, :] -= Q[r, 0:m] * H   (H = Hbuf, m x B); RB rows per workgroup
__global__ __launch_bounds__(256) void k_update_block(
    const double* __restrict__ Q, int ldq, int m, const double* __restrict__ Hbuf,
    double* __restrict__ W, int n) {
  __shared__ double Hl[kLdq * B];
  __shared__ double Ql[RB][kLdq + 1];
  const int tid = threadIdx.x;
  const int jj = tid % B, rr = tid / B;
  const int r0 = blockIdx.x * RB;
  for (int e = tid; e < m * B; e += 256) Hl[e] = Hbuf[e];
  for (int e = tid; e < RB * m; e += 256) {
    const int a = e / m, i = e - a * m;
    Ql[a][i] = (r0 + a < n) ? Q[(size_t)(r0 + a) * ldq + i] : 0.0;
  }
  __syncthreads();
  const int r = r0 + rr;
  if (r < n) {
    double acc = W[(size_t)r * B + jj];
    for (int i = 0; i < m; ++i) acc = __builtin_fma(-Ql[rr][i], Hl[i * B + jj], acc);
    W[(size_t)r * B + jj] = acc;
  }
}

__device__ __forceinline__ void chol8_wave(const double* G, double* __restrict__ Rinv,
                                           const double* __restrict__ hsq,
                                           int* __restrict__ flags,
                                           int* __restrict__ defect_flag, int flag_mode,
                                           int* __restrict__ sticky_flag = nullptr,
                                           int sticky_m = 0, bool sticky_on_defect = true);

// Gram reduce + Cholesky G = R^T R (right-looking, 256 threads), Rinv = R^-1 (upper).
// A column whose pivot is <= 1e-22 * (its own squared norm + what projection removed,
// hsq) is linearly dependent at working precision: it is zeroed and flagged for a
// random refill.
__global__ __launch_bounds__(256) void k_reduce_chol(
    const double* __restrict__ partial, int nparts, double* __restrict__ Rinv,
    double* __restrict__ Gsave, const double* __restrict__ hsq,
    int* __restrict__ flags, int* __restrict__ defect_flag, int flag_mode) {
  static_assert(B == 8, "the factorisation below maps the 8 x 8 matrix onto one wavefront");
  __shared__ double G[B * B];
  const int tid = threadIdx.x;
  {
    // Gram entry e summed by (256 / (B*B)) lanes, fixed-order butterfly
    constexpr int kLanes = 256 / (B * B);
    const int e = tid / kLanes, sub = tid % kLanes;
    double g = 0.0;
    for (int q = sub; q < nparts; q += kLanes) g += partial[(size_t)q * (kLdq * B) + e];
#pragma unroll
    for (int o = 1; o < kLanes; o <<= 1) g += __shfl_xor(g, o);
    if (sub == 0) {
      G[e] = g;
      if (Gsave) Gsave[e] = g;
    }
  }
  __syncthreads();
  if (tid >= 64) return;
  chol8_wave(G, Rinv, hsq, flags, defect_flag, flag_mode);
}

// 8 x 8 Cholesky G = R^T R + Rinv = R^-1 on ONE wavefront (the first 64 threads of the
// workgroup): lane (i, j) = (tid / 8, tid % 8) owns entry (i, j); the right-looking
// factorisation and the triangular inverse exchange rows / columns through shuffles, so the
// whole thing runs without a barrier.  G in LDS (or global), Rinv to global.
__device__ __forceinline__ void chol8_wave(const double* G, double* __restrict__ Rinv,
                                           const double* __restrict__ hsq,
                                           int* __restrict__ flags,
                                           int* __restrict__ defect_flag, int flag_mode,
                                           int* __restrict__ sticky_flag, int sticky_m,
                                           bool sticky_on_defect) {
  const int tid = threadIdx.x;
  const int i = tid >> 3, j = tid & 7;
  double g = G[tid];
  const double gii = __shfl(g, i * 9), gjj = __shfl(g, j * 9);  // original diagonal
  double defect = 0.0;
  if (flag_mode == 2 && gii != 0.0 && gjj != 0.0) defect = fabs(g - (i == j ? 1.0 : 0.0));
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) defect = fmax(defect, __shfl_xor(defect, o));
  double r = 0.0;  // R[i][j]
  int mask = 0;
  double pmax = 0.0, pmin = __builtin_huge_val();
#pragma unroll
  for (int k = 0; k < B; ++k) {
    const double d = __shfl(g, k * 9);
    const double total = __shfl(gii, k * 8) + (hsq ? hsq[k] : 0.0);
    const bool drop = !(d > 1e-22 * total) || !(d > 0.0);
    const double rk = drop ? 0.0 : sqrt(d);
    if (drop) {
      mask |= 1 << k;
    } else {
      pmax = fmax(pmax, rk);
      pmin = fmin(pmin, rk);
    }
    double row = 0.0;  // R[k][j] on the lanes of row k
    if (i == k && !drop) row = j == k ? rk : (j > k ? g / rk : 0.0);
    if (i == k) r = row;
    const double rki = __shfl(row, k * 8 + i), rkj = __shfl(row, k * 8 + j);
    if (i > k && j > k) g -= rki * rkj;
  }
  // R^-1 by back substitution, rows from the bottom up (dropped columns stay zero)
  const double rii = __shfl(r, i * 9), rjj = __shfl(r, j * 9);
  double ri = 0.0;  // Rinv[i][j]
#pragma unroll
  for (int srow = B - 1; srow >= 0; --srow) {
    double v = 0.0;
#pragma unroll
    for (int k = srow + 1; k < B; ++k) {
      const double rsk = __shfl(r, srow * 8 + k), rkj = __shfl(ri, k * 8 + j);
      if (k <= j) v -= rsk * rkj;
    }
    if (i == srow && j >= i && rii != 0.0 && rjj != 0.0)
      ri = i == j ? 1.0 / rii : v / rii;
  }
  Rinv[tid] = ri;
  if (tid == 0 && flags != nullptr) {
    flags[0] = mask;
    // An ill-conditioned block (pivot ratio > 1e3: the operator is numerically low-rank)
    // gets its small columns from R^-1 entries of that size, which amplify the absolute
    // Q-orthogonality error of the large columns: the caller must project against the
    // basis once more (twice-is-enough holds for the normalised block only).  flag_mode 2
    // (second CholQR pass): a Gram matrix far from I means the first pass met a block of
    // condition > 1e8.
    if (defect_flag != nullptr) *defect_flag = (defect > 0.1 || pmax > 1e3 * pmin) ? 1 : 0;
    // fused step chain (k_lz_step): nobody reads the flags between the blocks, so anything
    // that needs the careful host-driven path (a dependent column, a first CholQR pass that
    // met a block of condition > 1e8) is latched
    // ([1]: the basis size m of the first link that latched -- T[0:m, 0:m], the residual
    //  Gram and Q[:, 0:m] do not depend on this Cholesky and are still good; [2]: why)
    if (sticky_flag != nullptr && (mask != 0 || (sticky_on_defect && defect > 0.1))) {
      if (*sticky_flag == 0) {
        sticky_flag[1] = sticky_m;
        sticky_flag[2] = flag_mode * 1000 + mask;
      }
      *sticky_flag = 1;
    }
  }
}

// W <- W * Rinv; optional copies: Qdst[:, col0 + j] and Vs = c .* W; RB rows/workgroup
__global__ __launch_bounds__(256) void k_apply_rinv(
    double* __restrict__ W, int n, const double* __restrict__ Rinv,
    double* __restrict__ Qdst, int ldq, int col0, const double* __restrict__ cvec,
    double* __restrict__ Vs) {
  __shared__ double Rl[B * B];
  const int tid = threadIdx.x;
  if (tid < B * B) Rl[tid] = Rinv[tid];
  __syncthreads();
  const int jj = tid % B;
  const int r = blockIdx.x * RB + tid / B;
  const double w = r < n ? W[(size_t)r * B + jj] : 0.0;
  double v = 0.0;
#pragma unroll
  for (int k = 0; k < B; ++k) {
    const double wk = __shfl(w, k, B);
    v = __builtin_fma(wk, Rl[k * B + jj], v);
  }
  if (r < n) {
    W[(size_t)r * B + jj] = v;
    if (Qdst) Qdst[(size_t)r * ldq + col0 + jj] = v;
    if (Vs) Vs[(size_t)r * B + jj] = cvec[r] * v;
  }
}

// ---------------------------------------------------------------- block step chain
// The orthonormalisation of a Lanczos block as a chain of SHORT kernels, each about one
// global load round trip long (the tall-skinny work is latency-bound: a dependent kernel
// boundary costs ~1.5 us, a dependent miss ~1 us, an in-kernel cross-workgroup hand-off 5 us
// and more -- so links are separate launches and every link issues its loads in batches).
//
// One link = k_lz_rows, ROWS rows per workgroup:
//   prologue   every workgroup adds the partial sums the PREVIOUS link left (all workgroups,
//              fixed order: deterministic and identical everywhere) and turns them into this
//              link's coefficients -- the launch-boundary reduce, no separate kernel:
//     pre 1  Hc = Q^T W                                                          (CGS 1)
//     pre 2  Hc = Q^T W, G' = W^T W - Hc^T Hc = Gram of the projected block without another
//            pass (Pythagoras; Hc is rounding-level on a second projection), Rc = chol(G')^-1
//     pre 3  like 2 on the block the first Cholesky normalised: a third projection on a
//            well-conditioned block -- what the host-driven path does when a block was
//            ill-conditioned (pivot ratio > 1e3) -- at no extra pass here
//     pre 4  Rc = chol(W^T W)^-1                                           (start block)
//            workgroup 0 also records what the host side needs: T[:, col0 ..] (pre 1: set,
//            pre 2: accumulated), the column energies hsq, the residual Gram (pre 2), the
//            rank / conditioning flags.
//   body       W <- (W - Q[:, 0:m] Hc) Rc, optional store of the finished block
//              (Q[:, store_col ..], Vs = c .* W)
//   epilogue   this workgroup's share of the next link's sums, Q[:, 0:m]^T W and W^T W.
// A block costs matvec + 4 short launches and no host synchronisation; rank deficiency or a
// hopeless first Cholesky is latched in flags[13] and the caller falls back to the
// host-driven chain (orthonormalize / finish_block in eig_driver.hip).
constexpr int kLzThreads = 512;
constexpr int kLzPartStride = (kLdq + B) * B;  // proj rows [0, kLdq), Gram rows after
struct LzStep {
  double* W;
  int n;
  const double* Q;
  int ldq, m;
  // ---- prologue: reduce the previous link's partials
  int pre;                  // 0 none, else the mode above
  const double* pre_partial;
  int pre_nparts;
  double* T;
  int ldt, col0;
  double* Gsave;
  double* hsq;
  int* flags;
  int arm_defect;           // a Gram matrix far from I at this link's Cholesky latches flags[13]
  double* Tzero;            // start block: workgroup 0 clears T (kLdq x kLdq)
  // ---- body
  double* Qdst;             // store target (same buffer as Q, other columns) or nullptr
  int store_col;
  const double* vs_scale;
  double* Vs;
  int init_random;
  uint64_t seed;
  // ---- epilogue
  int want_proj, want_gram;
  double* partial;
};

// ordered sum of src[p * kLzPartStride], p < nparts; loads in independent batches of 16
__device__ __forceinline__ double lz_ordered_sum(const double* src, int nparts) {
  double acc = 0.0;
  for (int p0 = 0; p0 < nparts; p0 += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      v[u] = (p0 + u < nparts) ? src[(size_t)(p0 + u) * kLzPartStride] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  return acc;
}

template <int ROWS>
__device__ __forceinline__ void lz_rows_body(const LzStep& a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int kPass = kLzThreads / B;        // rows handled per sweep of the workgroup
  const int m = a.m;
  const int mq = m + 1;                        // row pitch of the basis rows in LDS
  double* Ql = smem;                           // ROWS x mq
  double* Hl = Ql + ROWS * mq;                 // m x B
  double* Rl = Hl + kLdq * B;                  // B x B
  double* Wl = Rl + B * B;                     // ROWS x B
  double* Gs = Wl + ROWS * B;                  // B x B
  double* hs = Gs + B * B;                     // B
  const int tid = threadIdx.x;
  const int j = tid % B;
  const int r0 = blockIdx.x * ROWS;
  const bool first_wg = blockIdx.x == 0;
  const bool pre_proj = (a.pre >= 1 && a.pre <= 3) && m > 0;
  const bool pre_gram = a.pre >= 2;
  const bool use_h = pre_proj;                 // apply Hc
  const bool use_r = pre_gram;                 // apply Rc
  const bool use_q = m > 0 && (use_h || a.want_proj);
  // ---- everything this workgroup reads, issued together: the previous link's partials,
  //      what workgroup 0 will read-modify-write, the basis rows, the block rows
  const int nproj = pre_proj ? m * B : 0;
  double hsum[3] = {0.0, 0.0, 0.0};            // up to (kLdq * B + 64) / 512 entries per thread
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int e = tid + u * kLzThreads;
    if (e < nproj) hsum[u] = lz_ordered_sum(a.pre_partial + e, a.pre_nparts);
    else if (pre_gram && e < nproj + B * B)
      hsum[u] = lz_ordered_sum(a.pre_partial + kLdq * B + (e - nproj), a.pre_nparts);
  }
  double told[3] = {0.0, 0.0, 0.0};
  double hsq_old = 0.0;
  if (first_wg && a.pre == 2) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + u * kLzThreads;
      if (e < nproj) told[u] = a.T[(size_t)(e / B) * a.ldt + a.col0 + (e % B)];
    }
    if (tid < B) hsq_old = a.hsq[tid];
  }
  if (!first_wg && a.pre == 2 && tid < B) hsq_old = a.hsq[tid];  // every Cholesky needs it
  if (use_q)
    for (int e = tid; e < ROWS * m; e += kLzThreads) {
      const int rr = e / m, i = e - rr * m;
      Ql[rr * mq + i] = (r0 + rr < a.n) ? a.Q[(size_t)(r0 + rr) * a.ldq + i] : 0.0;
    }
  double w[ROWS / kPass];
#pragma unroll
  for (int q = 0; q < ROWS / kPass; ++q) {
    const int r = r0 + q * kPass + tid / B;
    w[q] = 0.0;
    if (r < a.n)
      w[q] = a.init_random ? hash_uniform(a.seed, (uint64_t)r * B + j) : a.W[(size_t)r * B + j];
  }
  // ---- coefficients of this link
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int e = tid + u * kLzThreads;
    if (e < nproj) Hl[e] = hsum[u];
    else if (pre_gram && e < nproj + B * B) Gs[e - nproj] = hsum[u];
  }
  if (tid < B * B) Rl[tid] = (tid / B == tid % B) ? 1.0 : 0.0;
  __syncthreads();
  if (pre_proj && a.pre != 3) {
    if (first_wg && a.T != nullptr) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int e = tid + u * kLzThreads;
        if (e < nproj) {
          const int i = e / B, jc = a.col0 + (e % B);
          if (i <= jc) {
            const double t = a.pre == 2 ? told[u] + hsum[u] : hsum[u];
            a.T[(size_t)i * a.ldt + jc] = t;
            a.T[(size_t)jc * a.ldt + i] = t;
          }
        }
      }
    }
    if (tid < B) {  // column energies removed by projection
      double sq = 0.0;
      for (int i = 0; i < m; ++i) sq = __builtin_fma(Hl[i * B + tid], Hl[i * B + tid], sq);
      sq = a.pre == 2 ? hsq_old + sq : sq;
      // (global copy only from the CGS-1 link: in the CGS-2 link the other workgroups are
      //  still reading it)
      if (first_wg && a.pre == 1) a.hsq[tid] = sq;
      hs[tid] = sq;
    }
  } else if (tid < B) {
    hs[tid] = 0.0;
  }
  if (first_wg && a.Tzero != nullptr)
    for (int e = tid; e < kLdq * kLdq; e += kLzThreads) a.Tzero[e] = 0.0;
  if (pre_gram) {
    if (pre_proj && tid < B * B) {  // Gram of the projected block (Pythagoras)
      const int a1 = tid / B, b1 = tid % B;
      double g = Gs[tid];
      for (int i = 0; i < m; ++i) g = __builtin_fma(-Hl[i * B + a1], Hl[i * B + b1], g);
      Gs[tid] = g;
    }
    __syncthreads();
    if (first_wg && a.pre == 2 && a.Gsave != nullptr && tid < B * B) a.Gsave[tid] = Gs[tid];
    if (tid < 64)
      chol8_wave(Gs, Rl, a.pre == 2 ? hs : nullptr, first_wg ? a.flags : nullptr,
                 first_wg ? a.flags + (a.pre == 3 ? 10 : 11) : nullptr, a.pre == 3 ? 2 : 1,
                 first_wg ? a.flags + 13 : nullptr, m, a.arm_defect != 0);
    __syncthreads();
  }
  // ---- body
#pragma unroll
  for (int q = 0; q < ROWS / kPass; ++q) {
    const int rl = q * kPass + tid / B;
    const int r = r0 + rl;
    double x = w[q];
    if (use_h)
      for (int i = 0; i < m; ++i) x = __builtin_fma(-Ql[rl * mq + i], Hl[i * B + j], x);
    double v = x;
    if (use_r) {
      v = 0.0;
#pragma unroll
      for (int k = 0; k < B; ++k) v = __builtin_fma(__shfl(x, k, B), Rl[k * B + j], v);
    }
    if (r < a.n) {
      a.W[(size_t)r * B + j] = v;
      if (a.Qdst) a.Qdst[(size_t)r * a.ldq + a.store_col + j] = v;
      if (a.Vs) a.Vs[(size_t)r * B + j] = a.vs_scale[r] * v;
    }
    Wl[rl * B + j] = (r < a.n) ? v : 0.0;
  }
  if (!a.want_proj && !a.want_gram) return;
  __syncthreads();
  // ---- epilogue: partial sums over this workgroup's rows.  Every entry is split over two
  //      threads (lower / upper half of the rows) that are added lower + upper: fixed order.
  const int oproj = a.want_proj ? m * B : 0;
  const int nent = oproj + (a.want_gram ? B * B : 0);
  double* mine = a.partial + (size_t)blockIdx.x * kLzPartStride;
  for (int base = 0; base < nent; base += kLzThreads / 2) {
    const int e = base + (tid >> 1), half = tid & 1;
    double acc = 0.0;
    if (e < nent) {
      const int rbeg = half * (ROWS / 2);
      if (e < oproj) {
        const int i = e / B, jj = e % B;
#pragma unroll 8
        for (int rr = 0; rr < ROWS / 2; ++rr)
          acc = __builtin_fma(Ql[(rbeg + rr) * mq + i], Wl[(rbeg + rr) * B + jj], acc);
      } else {
        const int g = e - oproj, a1 = g / B, b1 = g % B;
#pragma unroll 8
        for (int rr = 0; rr < ROWS / 2; ++rr)
          acc = __builtin_fma(Wl[(rbeg + rr) * B + a1], Wl[(rbeg + rr) * B + b1], acc);
      }
    }
    const double other = __shfl_xor(acc, 1);
    if (e < nent && half == 0)
      mine[e < oproj ? e : kLdq * B + (e - oproj)] = acc + other;
  }
}
template <int ROWS>
__global__ __launch_bounds__(kLzThreads) void k_lz_rows(const LzStep a) {
  lz_rows_body<ROWS>(a);
}
template <int ROWS>
__global__ __launch_bounds__(kLzThreads) void k_lz_rows_g(const GroupOf<LzStep> g) {
  const LzStep& a = g.s[blockIdx.y];
  if ((int)blockIdx.x * ROWS >= a.n) return;
  lz_rows_body<ROWS>(a);
}

// fp64 reciprocal / reciprocal-sqrt from the hardware seeds (v_rcp_f64 / v_rsq_f64)
// plus two Newton steps each: ~1 ulp, a few dozen cycles instead of the ~250-cycle
// correctly-rounded sequences -- the rotation parameters are the serial part of
// every Jacobi round.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = __builtin_fma(y, 0.5 * __builtin_fma(-x * y, y, 1.0), y);
  y = __builtin_fma(y, 0.5 * __builtin_fma(-x * y, y, 1.0), y);
  return y;
}

// ---------------------------------------------------------------- dense Jacobi
// One workgroup, matrix A (mp x mp, mp = m rounded up to even) in LDS with row
// stride mp + 1; eigenvector accumulator kept TRANSPOSED in global memory
// (Yt(p)[i] = component i of vector p) so that the rotation of two vectors is
// two coalesced row updates.  Round-robin ordering: mp/2 disjoint rotations per
// round, mp - 1 rounds per sweep.
// mode 0: A = src (m x m).   mode 1: A_ij = c_i c_j src_ij + delta_ij p_i.
template <bool YT_LDS>
__global__ __launch_bounds__(1024) void k_jacobi(
    const double* __restrict__ src, int ldsrc, int m, int mode,
    const double* __restrict__ cvec, const double* __restrict__ pvec,
    const double* __restrict__ G, double* __restrict__ theta,
    double* __restrict__ Y, int ldy, double* __restrict__ resid,
    double* __restrict__ Yt_global, int* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const long long t_begin = wall_clock64();
  const long long c_begin = clock64();
  const int mp = (m + 1) & ~1;
  const int lda = mp + 1;
  double* A = smem;                       // mp * lda
  double* cs = A + mp * lda;              // mp/2
  double* sn = cs + mp / 2;               // mp/2
  int* pp = reinterpret_cast<int*>(sn + mp / 2);  // mp/2
  int* qq = pp + mp / 2;                  // mp/2
  // vector accumulator: LDS when it fits beside A (m <= 96), else global (L2).  The
  // choice is a template parameter: a runtime select would make every access a slow
  // generic (flat) one.
  double* Yl = sn + mp / 2 + mp / 2 + 2;
#define Yt(idx) (*(YT_LDS ? &Yl[idx] : &Yt_global[idx]))
  __shared__ int s_rot;
  const int tid = threadIdx.x;
  const int nth = blockDim.x;

  for (int e = tid; e < mp * mp; e += nth) {
    const int i = e / mp, j = e - i * mp;
    double v = 0.0;
    if (i < m && j < m) {
      v = src[(size_t)i * ldsrc + j];
      if (mode == 1) {
        v = cvec[i] * v * cvec[j];
        if (i == j) v += pvec[i];
      }
    }
    A[i * lda + j] = v;
    Yt((size_t)i * mp + j) = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  if (mode == 1) {  // enforce exact symmetry of the materialised operator
    for (int e = tid; e < mp * mp; e += nth) {
      const int i = e / mp, j = e - i * mp;
      if (i < j) {
        const double v = 0.5 * (A[i * lda + j] + A[j * lda + i]);
        A[i * lda + j] = v;
        A[j * lda + i] = v;
      }
    }
    __syncthreads();
  }

  // scale of the matrix (max |diagonal|): rotations whose off-diagonal entry is below
  // 1e-17 * scale are rounding noise -- without this floor a near-zero diagonal entry
  // (the Laplacian's null eigenvalue) keeps the relative test firing sweep after sweep
  __shared__ double s_scale;
  if (tid == 0) s_scale = 0.0;
  __syncthreads();
  if (tid < 64) {
    double mx = 0.0;
    for (int i = tid; i < m; i += 64) mx = fmax(mx, fabs(A[i * lda + i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if (tid == 0) s_scale = mx;
  }
  __syncthreads();
  const double tiny_abs = 1e-17 * s_scale, large_abs = 1e-13 * s_scale;
  const int half = mp / 2;
  int sweeps_done = 0;
  long long cyc_param = 0, cyc_bar1 = 0, cyc_upd = 0, cyc_bar2 = 0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    ++sweeps_done;
    if (tid == 0) s_rot = 0;
    __syncthreads();
    for (int round = 0; round < mp - 1; ++round) {
      const long long tr0 = clock64();
      // --- rotation parameters of the mp/2 disjoint pairs of this round.  Pair k is
      //     (p, q) = (round + k, round - k) mod (mp - 1) (k = 0: (mp - 1, round)); p
      //     and q are NOT sorted so that consecutive lanes touch consecutive columns.
      if (tid < half) {
        int p, q;
        if (tid == 0) {
          p = mp - 1;
          q = round;
        } else {
          p = round + tid;
          p = p >= mp - 1 ? p - (mp - 1) : p;
          q = round - tid;
          q = q < 0 ? q + (mp - 1) : q;
        }
        const double app = A[p * lda + p], aqq = A[q * lda + q];
        const double apq = A[p * lda + q];
        double c = 1.0, s = 0.0;
        const double apq2 = apq * apq;
        const double dd = fabs(app) * fabs(aqq);
        // rotate iff |apq| > 1e-18 sqrt(|app aqq|)   (compared squared: no sqrt)
        if (apq2 > 1e-36 * dd && fabs(apq) > tiny_abs && fabs(apq) > 1e-300) {
          // t = tan(angle): smaller root of t^2 + 2 theta t - 1 = 0, theta =
          // (aqq - app) / (2 apq), written without forming theta
          const double al = 0.5 * (aqq - app);
          const double h2 = __builtin_fma(al, al, apq2);
          double hh = h2 * fast_rsqrt(h2);                       // sqrt(al^2 + apq^2)
          hh = __builtin_fma(__builtin_fma(-hh, hh, h2), 0.5 * fast_rcp(hh), hh);
          const double den = al + copysign(hh, al);
          const double rd = fast_rcp(den);
          double t = apq * rd;
          t = __builtin_fma(__builtin_fma(-den, t, apq), rd, t);
          c = fast_rsqrt(__builtin_fma(t, t, 1.0));
          s = t * c;
          if (s != 0.0 && apq2 > 1e-18 * dd && fabs(apq) > large_abs)
            atomicAdd(&s_rot, 1);  // still "large"
        }
        cs[tid] = c;
        sn[tid] = s;
        pp[tid] = p;
        qq[tid] = q;
      }
      const long long tr1 = clock64();
      __syncthreads();
      const long long tr2 = clock64();
      // --- A <- J^T A J: each thread owns whole 2x2 blocks (pair k1 rows, pair k2
      //     columns): reads its 4 entries, writes its 4 entries, no other thread
      //     touches them, so one phase suffices.
      {
        const int lane = tid & 63, wave = tid >> 6, nwaves = nth >> 6;
        // --- vector part first: issue the loads of this thread's Yt items (rows p, q of
        //     up to kYtMax pairs) so their LDS latency overlaps the A update below
        // (static unroll 4 pairs x 2 lane steps covers mp <= 128 with the launcher's
        //  wave count; indices must be compile-time or the arrays land in scratch)
        double yp[8], yq[8];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = wave + kk * nwaves;
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = lane + 64 * ii;
            if (k < half && i < mp) {
              yp[kk * 2 + ii] = Yt((size_t)pp[k] * mp + i);
              yq[kk * 2 + ii] = Yt((size_t)qq[k] * mp + i);
            }
          }
        }
        // --- A <- J^T A J over 2x2 blocks; two pair-rows per wave when half <= 32
        const int rows_per_wave = half <= 32 ? 2 : 1;
        const int sub = rows_per_wave == 2 ? lane >> 5 : 0;
        const int l2 = rows_per_wave == 2 ? lane & 31 : lane;
        const int lstep = rows_per_wave == 2 ? 32 : 64;
        for (int k1 = wave * rows_per_wave + sub; k1 < half; k1 += nwaves * rows_per_wave) {
          const double c1 = cs[k1], s1 = sn[k1];
          const int p1 = pp[k1], q1 = qq[k1];
          for (int k2 = l2; k2 < half; k2 += lstep) {
            const double c2 = cs[k2], s2 = sn[k2];
            if (s1 == 0.0 && s2 == 0.0) continue;
            const int p2 = pp[k2], q2 = qq[k2];
            const double app = A[p1 * lda + p2], apq = A[p1 * lda + q2];
            const double aqp = A[q1 * lda + p2], aqq = A[q1 * lda + q2];
            // rows: J1^T
            const double tpp = c1 * app - s1 * aqp, tpq = c1 * apq - s1 * aqq;
            const double tqp = s1 * app + c1 * aqp, tqq = s1 * apq + c1 * aqq;
            // columns: J2
            double npp = c2 * tpp - s2 * tpq, npq = s2 * tpp + c2 * tpq;
            double nqp = c2 * tqp - s2 * tqq, nqq = s2 * tqp + c2 * tqq;
            if (k1 == k2) { npq = 0.0; nqp = 0.0; }  // the annihilated pair, exactly
            A[p1 * lda + p2] = npp;
            A[p1 * lda + q2] = npq;
            A[q1 * lda + p2] = nqp;
            A[q1 * lda + q2] = nqq;
          }
        }
        // --- finish the vectors
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = wave + kk * nwaves;
          if (k < half) {
            const double c = cs[k], s = sn[k];
            const int p = pp[k], q = qq[k];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
              const int i = lane + 64 * ii;
              if (i < mp && s != 0.0) {
                const double a = yp[kk * 2 + ii], b = yq[kk * 2 + ii];
                Yt((size_t)p * mp + i) = c * a - s * b;
                Yt((size_t)q * mp + i) = s * a + c * b;
              }
            }
          }
        }
      }
      const long long tr3 = clock64();
      __syncthreads();
      if (tid == 0) { cyc_param += tr1 - tr0; cyc_bar1 += tr2 - tr1; cyc_upd += tr3 - tr2; cyc_bar2 += clock64() - tr3; }
    }
    // s_rot counts rotations that were still "large" (|apq| > 1e-9 sqrt(app aqq)).
    // A sweep made only of small rotations leaves off-diagonals ~1e-18 relative
    // (quadratic convergence): done, without a further all-idle checking sweep.
    if (s_rot == 0) break;
    __syncthreads();
  }

  if (tid == 0 && dbg != nullptr) {
    dbg[1] = sweeps_done;
    dbg[2] = (int)(wall_clock64() - t_begin);  // 10 ns ticks
    dbg[3] = (int)((clock64() - c_begin) >> 10);  // shader cycles / 1024
    dbg[4] = (int)(cyc_param >> 10); dbg[5] = (int)(cyc_bar1 >> 10);
    dbg[6] = (int)(cyc_upd >> 10); dbg[7] = (int)(cyc_bar2 >> 10);
  }
  // --- sort descending, emit theta, Y (columns) and residual estimates
  for (int i = tid; i < m; i += nth) {
    const double di = A[i * lda + i];
    int rank = 0;
    for (int j = 0; j < m; ++j) {
      const double dj = A[j * lda + j];
      rank += (dj > di) || (dj == di && j < i);
    }
    theta[rank] = di;
    // residual estimate || Wres y_last ||, y_last = trailing 16 components
    if (resid) {
      double r2 = 0.0;
      if (G) {
        for (int a = 0; a < B; ++a) {
          const double ya = Yt((size_t)i * mp + (m - B + a));
          double t = 0.0;
          for (int b2 = 0; b2 < B; ++b2)
            t = __builtin_fma(G[a * B + b2], Yt((size_t)i * mp + (m - B + b2)), t);
          r2 = __builtin_fma(ya, t, r2);
        }
      }
      resid[rank] = sqrt(fmax(r2, 0.0));
    }
    // column `rank` of Y = vector i
    for (int r = 0; r < m; ++r) Y[(size_t)r * ldy + rank] = Yt((size_t)i * mp + r);
  }
}

#undef Yt

__global__ void k_set_diag_T(double* T, int ldt, int mtot, const double* theta,
                             int keep) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= mtot * mtot) return;
  const int i = e / mtot, j = e - i * mtot;
  T[(size_t)i * ldt + j] = (i == j && i < keep) ? theta[i] : 0.0;
}

// dst[r, 0:cols] = Q[r, 0:m] * Y[0:m, 0:cols]
__device__ __forceinline__ void basis_times_Y_body(
    const double* __restrict__ Q, int ldq, int m, const double* __restrict__ Y,
    int ldy, int cols, double* __restrict__ dst, int lddst, int n, int colmajor) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Yl = smem;                 // m x cols
  double* Ql = smem + m * cols;      // 16 x (m + 1)
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * 16;
  for (int e = tid; e < m * cols; e += 256) {
    const int i = e / cols, j = e - i * cols;
    Yl[e] = Y[(size_t)i * ldy + j];
  }
  for (int e = tid; e < 16 * m; e += 256) {
    const int a = e / m, i = e - a * m;
    Ql[a * (m + 1) + i] = (r0 + a < n) ? Q[(size_t)(r0 + a) * ldq + i] : 0.0;
  }
  __syncthreads();
  const int rr = tid >> 4;
  const int r = r0 + rr;
  if (r >= n) return;
  for (int j = tid & 15; j < cols; j += 16) {
    double acc = 0.0;
    for (int i = 0; i < m; ++i)
      acc = __builtin_fma(Ql[rr * (m + 1) + i], Yl[i * cols + j], acc);
    if (colmajor)
      dst[(size_t)j * lddst + r] = acc;  // one vector = one contiguous column
    else
      dst[(size_t)r * lddst + j] = acc;
  }
}
__global__ __launch_bounds__(256) void k_basis_times_Y(
    const double* __restrict__ Q, int ldq, int m, const double* __restrict__ Y,
    int ldy, int cols, double* __restrict__ dst, int lddst, int n, int colmajor) {
  basis_times_Y_body(Q, ldq, m, Y, ldy, cols, dst, lddst, n, colmajor);
}
__global__ __launch_bounds__(256) void k_basis_times_Y_g(const GroupOf<RitzItem> g) {
  const RitzItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x * 16 >= a.n) return;
  basis_times_Y_body(a.Q, a.ldq, a.m, a.Y, a.ldy, a.cols, a.E, a.lde, a.n, 1);
}

__global__ void k_copy_block(const double* __restrict__ src, int ldsrc,
                             double* __restrict__ dst, int lddst, int n, int cols) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * cols) return;
  const int r = e / cols, j = e - r * cols;
  dst[(size_t)r * lddst + j] = src[(size_t)r * ldsrc + j];
}

// Column-major eigenvectors: ET[j * ld + r].  One workgroup per column:
// v = t .* u, then v / ||v||_2  (LAPACK dgeev returns unit 2-norm columns).
// (blockDim.x = 256 or 1024 threads per column: a column of n = 8192 is two latency-bound
//  passes, 19 us with 256 threads)
__device__ __forceinline__ void back_transform_body(double* __restrict__ ET, int ld, int n,
                                                    const double* __restrict__ tvec) {
  __shared__ double sm[16];
  double* col = ET + (size_t)blockIdx.x * ld;
  const int nthr = blockDim.x;
  double acc = 0.0;
  for (int r = threadIdx.x; r < n; r += nthr) {
    const double v = tvec[r] * col[r];
    col[r] = v;
    acc = __builtin_fma(v, v, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  double tot = 0.0;
  for (int w = 0; w < nthr / 64; w += 4) tot += (sm[w] + sm[w + 1]) + (sm[w + 2] + sm[w + 3]);
  const double inv = 1.0 / sqrt(tot);
  for (int r = threadIdx.x; r < n; r += nthr) col[r] *= inv;
}
__global__ __launch_bounds__(1024) void k_back_transform(double* __restrict__ ET, int ld,
                                                         int n,
                                                         const double* __restrict__ tvec) {
  back_transform_body(ET, ld, n, tvec);
}
__global__ __launch_bounds__(256) void k_back_transform_g(const GroupOf<RitzItem> g) {
  const RitzItem& a = g.s[blockIdx.y];
  if ((int)blockIdx.x >= a.cols || a.n <= 0) return;
  back_transform_body(a.E, a.lde, a.n, a.tvec);
}

// dst (column-major, ldd) <- src (row-major, lds), n rows x cols
__global__ void k_rowmajor_to_colmajor(const double* __restrict__ src, int lds, int n,
                                       int cols, double* __restrict__ dst, int ldd) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * cols) return;
  const int j = e / n, r = e - j * n;
  dst[(size_t)j * ldd + r] = src[(size_t)r * lds + j];
}
// dst (row-major, ldd) <- src (column-major, lds)
__global__ void k_colmajor_to_rowmajor(const double* __restrict__ src, int lds, int n,
                                       int cols, double* __restrict__ dst, int ldd) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * cols) return;
  const int r = e / cols, j = e - r * cols;
  dst[(size_t)r * ldd + j] = src[(size_t)j * lds + r];
}

// ---------------------------------------------------------------- launchers
// number of partial-sum workgroups for the tall-skinny products: ~64 rows each
int proj_blocks(int n) { return std::max(1, std::min(kProjBlocks, (n + 63) / 64)); }
void launch_random_block(hipStream_t s, double* W, int n, uint64_t seed) {
  hipLaunchKernelGGL(k_random_block, dim3((n * B + 255) / 256), dim3(256), 0, s, W, n,
                     seed);
}
void launch_refill_deficient(hipStream_t s, double* W, int n, const int* flags,
                             uint64_t seed) {
  hipLaunchKernelGGL(k_refill_deficient, dim3((n * B + 255) / 256), dim3(256), 0, s,
                     W, n, flags, seed);
}
void launch_block_matvec(hipStream_t s, const double* S, int ld, int n,
                         const double* cvec, const double* pvec, const double* V,
                         int ldv, const double* Vs, double* W) {
  hipLaunchKernelGGL(k_block_matvec, dim3((n + 15) / 16), dim3(64 * kMvWaves), 0, s, S,
                     ld, n, cvec, pvec, V, ldv, Vs, W);
}
// slabs of the symmetric form: 2 x (tiles of the upper triangle) x 128 x B doubles
size_t matvec_sym_workspace_doubles(int n) {
  const size_t nt = (size_t)(n + kSymTile - 1) / kSymTile;
  return 2 * (nt * (nt + 1) / 2) * kSymTile * B;
}
// Same result from the upper triangle of a SYMMETRIC S (tiles with column block >= row block;
// the strictly lower tiles are never read).  ws: matvec_sym_workspace_doubles(n).
void launch_block_matvec_sym(hipStream_t s, const double* S, int ld, int n, const double* cvec,
                             const double* pvec, const double* V, int ldv, const double* Vs,
                             double* W, double* ws) {
  const int nt = (n + kSymTile - 1) / kSymTile;
  const int items = nt * (nt + 1) / 2;
  double* pd = ws;
  double* pm = ws + (size_t)items * kSymTile * B;
  hipLaunchKernelGGL(k_block_matvec_sym, dim3(items), dim3(256), 0, s, S, ld, n, Vs, pd, pm, nt);
  hipLaunchKernelGGL(k_matvec_sym_reduce, dim3(nt * 4), dim3(256), 0, s, pd, pm, nt, n, cvec,
                     pvec, V, ldv, W);
}
void launch_proj_partial(hipStream_t s, const double* Q, int ldq, int m,
                         const double* W, int n, double* partial) {
  hipLaunchKernelGGL(k_proj_partial, dim3(proj_blocks(n)), dim3(256), 0, s, Q, ldq, m, W,
                     n, partial);
}
void launch_reduce_H(hipStream_t s, const double* partial, int nparts, int m, double* Hbuf,
                     double* T, int ldt, int col0, int accumulate, double* hsq) {
  hipLaunchKernelGGL(k_reduce_H, dim3((m * B + 31) / 32), dim3(256), 0, s, partial,
                     nparts, m, Hbuf, T, ldt, col0, accumulate, hsq);
}
void launch_update_block(hipStream_t s, const double* Q, int ldq, int m,
                         const double* Hbuf, double* W, int n) {
  hipLaunchKernelGGL(k_update_block, dim3((n + RB - 1) / RB), dim3(256), 0, s, Q, ldq, m,
                     Hbuf, W, n);
}
void launch_reduce_chol(hipStream_t s, const double* partial, int nparts, double* Rinv,
                        double* Gsave, const double* hsq, int* flags, int* defect_flag,
                        int flag_mode) {
  hipLaunchKernelGGL(k_reduce_chol, dim3(1), dim3(256), 0, s, partial, nparts,
                     Rinv, Gsave, hsq, flags, defect_flag, flag_mode);
}
void launch_apply_rinv(hipStream_t s, double* W, int n, const double* Rinv,
                       double* Qdst, int ldq, int col0, const double* cvec,
                       double* Vs) {
  hipLaunchKernelGGL(k_apply_rinv, dim3((n + RB - 1) / RB), dim3(256), 0, s, W, n, Rinv,
                     Qdst, ldq, col0, cvec, Vs);
}
// rows per workgroup of k_lz_rows: as many as the basis rows leave room for in LDS (fewer,
// fatter workgroups = fewer partials for the next link to add)
static int lz_rows_for(int m) {
  // (128-row workgroups by default: twice the workgroups of the 256-row form, half the LDS
  //  fill and half the serial length of the epilogue sums per link -- eigen stage 0.536 ->
  //  0.492 ms at n = 8192; 64 rows: 0.499)
  if ((size_t)128 * (m + 1) * sizeof(double) <= 112 * 1024) return 128;
  if ((size_t)256 * (m + 1) * sizeof(double) <= 96 * 1024) return 256;
  if ((size_t)128 * (m + 1) * sizeof(double) <= 112 * 1024) return 128;
  return 64;
}
// two partial buffers: a link reads its predecessor's sums while it writes its own
size_t lz_partial_doubles(int n) { return 2 * (size_t)((n + 63) / 64) * kLzPartStride; }

template <int ROWS>
static void launch_rows(hipStream_t s, const LzStep& a, int nwg) {
  const size_t lds = sizeof(double) * ((size_t)ROWS * (a.m + 1) + (size_t)kLdq * B + B * B +
                                       (size_t)ROWS * B + B * B + B);
  static std::once_flag once[16];
  int dev = 0;
  hipGetDevice(&dev);
  std::call_once(once[dev & 15], [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_lz_rows<ROWS>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  });
  hipLaunchKernelGGL(k_lz_rows<ROWS>, dim3(nwg), dim3(kLzThreads), lds, s, a);
}

// One link of the chain (see k_lz_rows).  `pre`: what to make of the previous link's partial
// sums (0 nothing, 1 CGS-1, 2 CGS-2 + Cholesky, 3 re-projection + Cholesky, 4 Gram +
// Cholesky); `next`: which sums to leave for the next link (same codes, 0 = none).
// chain->parity / chain->nparts carry the partial buffer in use from link to link.
static LzStep lz_make_step(const EigWorkspace& ws, LzChain* chain, int n, int m, int pre,
                           int next, int store_col, const double* vs_scale, int col0,
                           bool init_random, uint64_t seed, bool zero_T) {
  LzStep a;
  // the three-pass form (chain->three_pass) re-orthonormalises once more: only its LAST
  // Cholesky (the link that stores the block) may still latch on a Gram matrix far from I
  a.arm_defect = (!chain->three_pass || store_col >= 0) ? 1 : 0;
  a.W = ws.W;
  a.n = n;
  a.Q = ws.Q;
  a.ldq = kLdq;
  a.m = m;
  a.pre = pre;
  const size_t half = lz_partial_doubles(n) / 2;
  a.pre_partial = ws.partial + (size_t)chain->parity * half;
  a.pre_nparts = chain->nparts;
  a.T = ws.T;
  a.ldt = kLdq;
  a.col0 = col0;
  a.Gsave = ws.G;
  a.hsq = ws.hsq;
  a.flags = ws.flags;
  a.Tzero = zero_T ? ws.T : nullptr;
  a.Qdst = store_col >= 0 ? ws.Q : nullptr;
  a.store_col = store_col >= 0 ? store_col : 0;
  a.vs_scale = vs_scale;
  a.Vs = store_col >= 0 ? ws.Vs : nullptr;
  a.init_random = init_random ? 1 : 0;
  a.seed = seed;
  a.want_proj = (next >= 1 && next <= 3 && m > 0) ? 1 : 0;
  a.want_gram = next >= 2 ? 1 : 0;
  chain->parity ^= 1;
  a.partial = ws.partial + (size_t)chain->parity * half;
  const int rows = lz_rows_for(m);
  chain->nparts = (n + rows - 1) / rows;
  return a;
}
void launch_lz_link(hipStream_t s, const EigWorkspace& ws, LzChain* chain, int n, int m,
                    int pre, int next, int store_col, const double* vs_scale, int col0,
                    bool init_random, uint64_t seed, bool zero_T) {
  const LzStep a = lz_make_step(ws, chain, n, m, pre, next, store_col, vs_scale, col0,
                                init_random, seed, zero_T);
  const int rows = lz_rows_for(m);
  const int nwg = chain->nparts;
  if (rows == 256) launch_rows<256>(s, a, nwg);
  else if (rows == 128) launch_rows<128>(s, a, nwg);
  else launch_rows<64>(s, a, nwg);
}

// What a Rayleigh-Ritz check reads, of every member, compacted into one buffer (one copy to
// the host for the whole group): T[0:m, 0:m] row-major with pitch m, the 8 x 8 residual Gram,
// the 16 flag words.  One workgroup per member; T == nullptr: idle.
__global__ __launch_bounds__(256) void k_group_gather(const GroupOf<GatherItem> g, int m,
                                                      double* __restrict__ out, int stride) {
  const GatherItem& a = g.s[blockIdx.x];
  if (a.T == nullptr) return;
  double* dst = out + (size_t)blockIdx.x * stride;
  for (int e = threadIdx.x; e < m * m; e += 256) {
    const int i = e / m, j = e - i * m;
    dst[e] = a.T[(size_t)i * kLdq + j];
  }
  if (threadIdx.x < B * B) dst[m * m + threadIdx.x] = a.G[threadIdx.x];
  if (threadIdx.x < 16) reinterpret_cast<int*>(dst + m * m + B * B)[threadIdx.x] = a.flags[threadIdx.x];
}
void launch_group_gather(hipStream_t s, const GatherItem* items, int count, int m, double* out,
                         int stride) {
  GroupOf<GatherItem> g;
  memset(&g, 0, sizeof(g));
  for (int z = 0; z < count; ++z) g.s[z] = items[z];
  hipLaunchKernelGGL(k_group_gather, dim3(count), dim3(256), 0, s, g, m, out, stride);
}

// ---- grouped launches: one link / matvec / Ritz-vector product for up to kGroupMax
//      independent problems that advance in lockstep (same m); idle members carry n = 0
template <int ROWS>
static void launch_rows_group(hipStream_t s, const GroupOf<LzStep>& g, int m, int nwg,
                              int count) {
  const size_t lds = sizeof(double) * ((size_t)ROWS * (m + 1) + (size_t)kLdq * B + B * B +
                                       (size_t)ROWS * B + B * B + B);
  SC_OPT_IN_LDS(k_lz_rows_g<ROWS>, 150 * 1024);
  hipLaunchKernelGGL(k_lz_rows_g<ROWS>, dim3(nwg, count), dim3(kLzThreads), lds, s, g);
}
void launch_lz_link_group(hipStream_t s, LzGroupMember* mem, int count, int m, int pre,
                          int next, int store_col, int col0, bool init_random, uint64_t seed,
                          bool zero_T) {
  GroupOf<LzStep> g;
  memset(&g, 0, sizeof(g));
  const int rows = lz_rows_for(m);
  int nwg = 0;
  for (int z = 0; z < count; ++z) {
    if (!mem[z].active) continue;  // n stays 0: every workgroup of the member returns
    g.s[z] = lz_make_step(mem[z].ws, &mem[z].chain, mem[z].n, m, pre, next, store_col,
                          mem[z].vs_scale, col0, init_random, seed, zero_T);
    mem[z].chain.nparts = (mem[z].n + rows - 1) / rows;  // (rows may be capped here)
    nwg = std::max(nwg, mem[z].chain.nparts);
  }
  if (nwg == 0) return;
  if (rows == 256) launch_rows_group<256>(s, g, m, nwg, count);
  else if (rows == 128) launch_rows_group<128>(s, g, m, nwg, count);
  else launch_rows_group<64>(s, g, m, nwg, count);
}
// `symmetric`: from the upper triangles only (k_block_matvec_sym + its reduce; every member
// needs its slab workspace, MatvecItem::slabs).  The matrices of a group rarely fit the
// caches together, so the halved traffic pays from much smaller n than for a single call.
void launch_block_matvec_group(hipStream_t s, const MatvecItem* items, int count,
                               bool symmetric) {
  GroupOf<MatvecItem> g;
  memset(&g, 0, sizeof(g));
  int nmax = 0;
  for (int z = 0; z < count; ++z) {
    g.s[z] = items[z];
    nmax = std::max(nmax, items[z].n);
  }
  if (nmax == 0) return;
  if (symmetric) {
    const int nt = (nmax + kSymTile - 1) / kSymTile;
    hipLaunchKernelGGL(k_block_matvec_sym_g, dim3(nt * (nt + 1) / 2, count), dim3(256), 0, s, g);
    hipLaunchKernelGGL(k_matvec_sym_reduce_g, dim3(nt * 4, count), dim3(256), 0, s, g);
    return;
  }
  hipLaunchKernelGGL(k_block_matvec_g, dim3((nmax + 15) / 16, count), dim3(64 * kMvWaves), 0,
                     s, g);
}
// E[:, 0:cols] = normalise(t .* (Q[:, 0:m] Y[0:m, 0:cols])) for every member (n = 0: idle)
void launch_ritz_vectors_group(hipStream_t s, const RitzItem* items, int count) {
  GroupOf<RitzItem> g;
  memset(&g, 0, sizeof(g));
  int nmax = 0, cmax = 0;
  size_t lds = 0;
  for (int z = 0; z < count; ++z) {
    g.s[z] = items[z];
    if (items[z].n <= 0) continue;
    nmax = std::max(nmax, items[z].n);
    cmax = std::max(cmax, items[z].cols);
    lds = std::max(lds, sizeof(double) * ((size_t)items[z].m * items[z].cols +
                                          16 * (size_t)(items[z].m + 1)));
  }
  if (nmax == 0) return;
  SC_OPT_IN_LDS(k_basis_times_Y_g, 128 * 1024);
  hipLaunchKernelGGL(k_basis_times_Y_g, dim3((nmax + 15) / 16, count), dim3(256), lds, s, g);
  hipLaunchKernelGGL(k_back_transform_g, dim3(cmax, count), dim3(256), 0, s, g);
}
void launch_jacobi(hipStream_t s, const double* src, int ldsrc, int m, int mode,
                   const double* cvec, const double* pvec, const double* G,
                   double* theta, double* Y, int ldy, double* resid, double* Yt, int* dbg) {
  const int mp = (m + 1) & ~1;
  const size_t base = sizeof(double) * ((size_t)mp * (mp + 1) + 2 * mp + 2) + 64;
  const size_t with_yt = base + sizeof(double) * (size_t)mp * mp;
  const int yt_in_lds = with_yt <= 150 * 1024;
  const size_t lds = yt_in_lds ? with_yt : base;
  SC_OPT_IN_LDS(k_jacobi<true>, 160 * 1024 - 256);
  SC_OPT_IN_LDS(k_jacobi<false>, 160 * 1024 - 256);
  // always 16 waves: each round is a chain of dependent LDS round trips, so the time
  // per round is set by how many work items a wave handles one after the other
  const int threads = 1024;
  if (yt_in_lds)
    hipLaunchKernelGGL(k_jacobi<true>, dim3(1), dim3(threads), lds, s, src, ldsrc, m, mode,
                       cvec, pvec, G, theta, Y, ldy, resid, Yt, dbg);
  else
    hipLaunchKernelGGL(k_jacobi<false>, dim3(1), dim3(threads), lds, s, src, ldsrc, m, mode,
                       cvec, pvec, G, theta, Y, ldy, resid, Yt, dbg);
}
void launch_set_diag_T(hipStream_t s, double* T, int ldt, int mtot,
                       const double* theta, int keep) {
  hipLaunchKernelGGL(k_set_diag_T, dim3((mtot * mtot + 255) / 256), dim3(256), 0, s, T,
                     ldt, mtot, theta, keep);
}
void launch_basis_times_Y(hipStream_t s, const double* Q, int ldq, int m,
                          const double* Y, int ldy, int cols, double* dst,
                          int lddst, int n, int colmajor) {
  const size_t lds = sizeof(double) * ((size_t)m * cols + 16 * (size_t)(m + 1));
  SC_OPT_IN_LDS(k_basis_times_Y, 128 * 1024);
  hipLaunchKernelGGL(k_basis_times_Y, dim3((n + 15) / 16), dim3(256), lds, s, Q, ldq,
                     m, Y, ldy, cols, dst, lddst, n, colmajor);
}
void launch_copy_block(hipStream_t s, const double* src, int ldsrc, double* dst,
                       int lddst, int n, int cols) {
  hipLaunchKernelGGL(k_copy_block, dim3((n * cols + 255) / 256), dim3(256), 0, s, src,
                     ldsrc, dst, lddst, n, cols);
}
void launch_back_transform(hipStream_t s, double* ET, int ld, int n, int cols,
                           const double* tvec) {
  hipLaunchKernelGGL(k_back_transform, dim3(cols), dim3(n >= 4096 ? 1024 : 256), 0, s, ET, ld, n,
                     tvec);
}
void launch_rowmajor_to_colmajor(hipStream_t s, const double* src, int lds, int n, int cols,
                                 double* dst, int ldd) {
  hipLaunchKernelGGL(k_rowmajor_to_colmajor, dim3((n * cols + 255) / 256), dim3(256), 0, s,
                     src, lds, n, cols, dst, ldd);
}
void launch_colmajor_to_rowmajor(hipStream_t s, const double* src, int lds, int n, int cols,
                                 double* dst, int ldd) {
  hipLaunchKernelGGL(k_colmajor_to_rowmajor, dim3((n * cols + 255) / 256), dim3(256), 0, s,
                     src, lds, n, cols, dst, ldd);
}

}  // namespace sc
