// fp64 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T  (both operands
// row-major with K contiguous, i.e. the "NT" form both hot products have):
//   * affinity  C = Xn Xn^T, epilogue (c + 1) / 2      (reference utils.py:35-39)
//   * Diffuse   C = A A^T                               (reference refinement.py:234)
// Both outputs are symmetric, so the SYM variant computes only tile pairs
// (ti <= tj) and writes the mirror tile transposed: half the flops, and the
// result is exactly symmetric (what numpy's syrk-backed A @ A.T gives).
//
// Tiling: 128x128 block tile, BK = 16, 256 threads = 4 waves, each wave a 64x64
// sub-tile = 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 acc VGPRs).
// LDS tiles are [128][16] doubles with the 16-byte chunk index XOR-swizzled by
// (row >> 1) & 7 so the per-lane ds_read_b64 of an MFMA operand (16 rows x 2 k
// per 32-lane group) hits 32 distinct bank pairs.
// Global loads are register-staged one K-tile ahead.
#include "sc_internal.h"

namespace sc {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 16;

__device__ __forceinline__ int lds_chunk_off(int row, int kc) {
  return row * BK + (((kc ^ ((row >> 1) & 7))) << 1);
}

template <int EPI, bool SYM>
__global__ __launch_bounds__(256) void k_gemm_nt(const double* __restrict__ A,
                                                 int lda,
                                                 const double* __restrict__ B,
                                                 int ldb, double* __restrict__ C,
                                                 int ldc, int M, int N, int K,
                                                 int ntiles) {
  __shared__ __attribute__((aligned(16))) double As[BM * BK];
  __shared__ __attribute__((aligned(16))) double Bs[BN * BK];

  int ti, tj;
  if (SYM) {
    // linear id over the upper triangle (ti <= tj), row by row
    int id = blockIdx.x;
    ti = 0;
    int rowlen = ntiles;
    while (id >= rowlen) {
      id -= rowlen;
      --rowlen;
      ++ti;
    }
    tj = ti + id;
  } else {
    ti = blockIdx.y;
    tj = blockIdx.x;
  }
  const int row0 = ti * BM;
  const int col0 = tj * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1;
  const int wc = wave & 1;
  const int li = lane & 15;
  const int lg = lane >> 4;

  // --- global -> register staging: 4 chunks (16 B) of A and of B per thread
  const double* aptr[4];
  const double* bptr[4];
  int lds_off[4];
  int kcol[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = tid + 256 * q;
    const int r = c >> 3;
    const int kc = c & 7;
    int ga = row0 + r;
    ga = ga < M ? ga : M - 1;
    int gb = col0 + r;
    gb = gb < N ? gb : N - 1;
    aptr[q] = A + (size_t)ga * lda + 2 * kc;
    bptr[q] = B + (size_t)gb * ldb + 2 * kc;
    lds_off[q] = lds_chunk_off(r, kc);
    kcol[q] = 2 * kc;
  }

  v4f64 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = (v4f64){0.0, 0.0, 0.0, 0.0};

  const int ktiles = (K + BK - 1) / BK;
  double2 ra[4], rb[4];

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double2 va = *reinterpret_cast<const double2*>(aptr[q] + k0);
      double2 vb = *reinterpret_cast<const double2*>(bptr[q] + k0);
      const int k = k0 + kcol[q];
      if (k >= K) { va.x = 0.0; vb.x = 0.0; }
      if (k + 1 >= K) { va.y = 0.0; vb.y = 0.0; }
      ra[q] = va;
      rb[q] = vb;
    }
  };

  gload(0);

  // operand read offsets inside a tile (row part); k part added per sub-step
  const int swz = (li >> 1) & 7;
  const int arow = (wr * 64 + li) * BK;
  const int brow = (wc * 64 + li) * BK;

  for (int kt = 0; kt < ktiles; ++kt) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<double2*>(&As[lds_off[q]]) = ra[q];
      *reinterpret_cast<double2*>(&Bs[lds_off[q]]) = rb[q];
    }
    __syncthreads();
    if (kt + 1 < ktiles) gload(kt + 1);

#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kc = 2 * s + (lg >> 1);
      const int koff = ((kc ^ swz) << 1) + (lg & 1);
      double a[4], b[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = As[arow + m * 16 * BK + koff];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) b[nn] = Bs[brow + nn * 16 * BK + koff];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
          acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[nn],
                                                            acc[m][nn], 0, 0, 0);
    }
  }

  // --- epilogue.  D layout of v_mfma_f64_16x16x4_f64: lane l, reg r holds
  //     D[row = (l >> 4) + 4 r][col = l & 15].
  const bool mirror = SYM && (ti != tj);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      const int col = col0 + wc * 64 + nn * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wr * 64 + m * 16 + lg + 4 * r;
        double v = acc[m][nn][r];
        if (EPI == kEpiAffinity) v = (v + 1.0) * 0.5;  // == (v + 1) / 2 exactly
        if (row < M && col < N) {
          C[(size_t)row * ldc + col] = v;
          if (mirror) C[(size_t)col * ldc + row] = v;
        }
      }
    }
  }
}

void launch_gemm_nt(hipStream_t s, const double* A, int lda, const double* B,
                    int ldb, double* C, int ldc, int M, int N, int K,
                    int epilogue, bool symmetric) {
  if (M <= 0 || N <= 0) return;
  const int tm = (M + BM - 1) / BM;
  const int tn = (N + BN - 1) / BN;
  if (symmetric) {
    dim3 grid(tm * (tm + 1) / 2);
    if (epilogue == kEpiAffinity)
      hipLaunchKernelGGL((k_gemm_nt<kEpiAffinity, true>), grid, dim3(256), 0, s, A,
                         lda, B, ldb, C, ldc, M, N, K, tm);
    else
      hipLaunchKernelGGL((k_gemm_nt<kEpiNone, true>), grid, dim3(256), 0, s, A,
                         lda, B, ldb, C, ldc, M, N, K, tm);
  } else {
    dim3 grid(tn, tm);
    if (epilogue == kEpiAffinity)
      hipLaunchKernelGGL((k_gemm_nt<kEpiAffinity, false>), grid, dim3(256), 0, s,
                         A, lda, B, ldb, C, ldc, M, N, K, tm);
    else
      hipLaunchKernelGGL((k_gemm_nt<kEpiNone, false>), grid, dim3(256), 0, s, A,
                         lda, B, ldb, C, ldc, M, N, K, tm);
  }
}

}  // namespace sc
