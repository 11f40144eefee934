// Symmetric top-k eigensolver kernels (reference utils.py:44-71 calls LAPACK
// dgeev on a matrix that is diagonally similar to a symmetric one).
//
// Operator:  Op = diag(p) + diag(c) S diag(c),  S = refined (symmetric) matrix.
// Method:    block Lanczos, block = 16 vectors (one v_mfma_f64_16x16x4_f64 tile
//            column), full re-orthogonalisation (CGS2 + CholQR2), explicit
//            Rayleigh-Ritz T = Q^T Op Q, thick restart; the small dense
//            eigenproblem is a one-workgroup cyclic Jacobi with T in LDS.
// The only O(n^2) kernel is k_block_matvec (one HBM pass over S per 16 vectors);
// everything else is tall-skinny (n x <=144) and L2-resident.
#include "sc_internal.h"

namespace sc {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int B = kEigBlock;  // 16

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
// One workgroup = 16 rows of S; its 4 waves split K in interleaved 32-wide
// chunks.  Lane (i = l & 15, g = l >> 4) loads S[r0 + i][kb + 8 g .. + 7] (64 B
// contiguous) and feeds it to 8 MFMAs whose k-slot g carries k = kb + 8 g + t.
__global__ __launch_bounds__(256) void k_block_matvec(
    const double* __restrict__ S, int ld, int n, const double* __restrict__ cvec,
    const double* __restrict__ pvec, const double* __restrict__ V, int ldv,
    const double* __restrict__ Vs, double* __restrict__ W) {
  __shared__ double red[4][64][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int r0 = blockIdx.x * 16;
  int ri = r0 + li;
  ri = ri < n ? ri : n - 1;
  const double* srow = S + (size_t)ri * ld;
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  const int nchunks = (n + 31) / 32;
  for (int ch = wave; ch < nchunks; ch += 4) {
    const int kb = ch * 32 + 8 * lg;
    double a[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // ld is a multiple of 16 and kb + 7 < ld, so the load stays in the row
      const double2 v = *reinterpret_cast<const double2*>(srow + kb + 2 * q);
      a[2 * q] = v.x;
      a[2 * q + 1] = v.y;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = kb + t;
      const bool ok = k < n;
      const double av = ok ? a[t] : 0.0;
      const double bv = ok ? Vs[(size_t)k * B + li] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double sum = (red[0][lane][r] + red[1][lane][r]) +
                         (red[2][lane][r] + red[3][lane][r]);
      const int row = r0 + lg + 4 * r;  // D[row = (l >> 4) + 4 r][col = l & 15]
      if (row < n)
        W[(size_t)row * B + li] =
            __builtin_fma(cvec[row], sum, pvec[row] * V[(size_t)row * ldv + li]);
    }
  }
}

// ---------------------------------------------------------------- projections
// partial[blk][i * 16 + j] = sum over the block's rows of Q[r][i] * W[r][j]
__global__ __launch_bounds__(256) void k_proj_partial(
    const double* __restrict__ Q, int ldq, int m, const double* __restrict__ W,
    int n, double* __restrict__ partial) {
  __shared__ double Ql[8][kLdq];
  __shared__ double Wl[8][B];
  const int tid = threadIdx.x;
  const int jj = tid & 15, ig = tid >> 4;
  const int rows_per = (n + gridDim.x - 1) / gridDim.x;
  const int rbeg = blockIdx.x * rows_per;
  const int rend = min(n, rbeg + rows_per);
  double acc[kLdq / 16];
#pragma unroll
  for (int q = 0; q < kLdq / 16; ++q) acc[q] = 0.0;
  for (int r = rbeg; r < rend; r += 8) {
    __syncthreads();
    for (int e = tid; e < 8 * m; e += 256) {
      const int rr = e / m, i = e - rr * m;
      Ql[rr][i] = (r + rr < rend) ? Q[(size_t)(r + rr) * ldq + i] : 0.0;
    }
    if (tid < 8 * B) {
      const int rr = tid >> 4;
      Wl[rr][jj] = (r + rr < rend) ? W[(size_t)(r + rr) * B + jj] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const double w = Wl[rr][jj];
#pragma unroll
      for (int q = 0; q < kLdq / 16; ++q)
        if (ig + 16 * q < m) acc[q] = __builtin_fma(Ql[rr][ig + 16 * q], w, acc[q]);
    }
  }
  double* out = partial + (size_t)blockIdx.x * (kLdq * B);
#pragma unroll
  for (int q = 0; q < kLdq / 16; ++q)
    if (ig + 16 * q < m) out[(ig + 16 * q) * B + jj] = acc[q];
}

// H = sum of partials (m x 16), stored to Hbuf; optionally accumulated into
// T[0:m, col0:col0+16] and mirrored (upper triangle of the diagonal block only);
// hsq[j] (+)= sum_i H_ij^2.
__global__ __launch_bounds__(256) void k_reduce_H(
    const double* __restrict__ partial, int nparts, int m, double* __restrict__ Hbuf,
    double* __restrict__ T, int ldt, int col0, int accumulate,
    double* __restrict__ hsq) {
  __shared__ double sq[256];
  const int tid = threadIdx.x;
  double mysq = 0.0;
  for (int e = tid; e < m * B; e += 256) {
    double h = 0.0;
    for (int g = 0; g < nparts; ++g) h += partial[(size_t)g * (kLdq * B) + e];
    Hbuf[e] = h;
    mysq = __builtin_fma(h, h, mysq);
    if (T != nullptr) {
      const int i = e >> 4, j = col0 + (e & 15);
      if (i <= j) {
        const double v = accumulate ? T[(size_t)i * ldt + j] + h : h;
        T[(size_t)i * ldt + j] = v;
        T[(size_t)j * ldt + i] = v;
      }
    }
  }
  // per-column sums of squares: thread tid owns column tid & 15 in every stride
  sq[tid] = mysq;
  __syncthreads();
  if (tid < B) {
    double s = 0.0;
    for (int q = tid; q < 256; q += B) s += sq[q];
    hsq[tid] = accumulate ? hsq[tid] + s : s;
  }
}

// W[r, :] -= Q[r, 0:m] * H   (H = Hbuf, m x 16)
__global__ __launch_bounds__(256) void k_update_block(
    const double* __restrict__ Q, int ldq, int m, const double* __restrict__ Hbuf,
    double* __restrict__ W, int n) {
  __shared__ double Hl[kLdq * B];
  __shared__ double Ql[16][kLdq + 1];
  const int tid = threadIdx.x;
  const int jj = tid & 15, rr = tid >> 4;
  const int r0 = blockIdx.x * 16;
  for (int e = tid; e < m * B; e += 256) Hl[e] = Hbuf[e];
  for (int e = tid; e < 16 * m; e += 256) {
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

// Gram reduce + Cholesky G = R^T R, Rinv = R^-1 (upper).  A column whose pivot is
// <= 1e-22 * (its own squared norm + what projection removed, hsq) is linearly
// dependent at working precision: it is zeroed and flagged for a random refill.
__global__ __launch_bounds__(256) void k_reduce_chol(
    const double* __restrict__ partial, int nparts, double* __restrict__ Rinv,
    double* __restrict__ Gsave, const double* __restrict__ hsq,
    int* __restrict__ flags) {
  __shared__ double G[B][B];
  __shared__ double R[B][B];
  __shared__ double Ri[B][B];
  const int tid = threadIdx.x;
  {
    double g = 0.0;
    for (int q = 0; q < nparts; ++q) g += partial[(size_t)q * (kLdq * B) + tid];
    G[tid >> 4][tid & 15] = g;
    if (Gsave) Gsave[tid] = g;
    R[tid >> 4][tid & 15] = 0.0;
    Ri[tid >> 4][tid & 15] = 0.0;
  }
  __syncthreads();
  if (tid == 0) {
    int mask = 0;
    for (int j = 0; j < B; ++j) {
      double d = G[j][j];
      for (int k = 0; k < j; ++k) d -= R[k][j] * R[k][j];
      const double total = G[j][j] + (hsq ? hsq[j] : 0.0);
      if (!(d > 1e-22 * total) || !(d > 0.0)) {
        mask |= 1 << j;
        R[j][j] = 0.0;  // marks a dropped column
        continue;
      }
      const double rjj = sqrt(d);
      R[j][j] = rjj;
      for (int c2 = j + 1; c2 < B; ++c2) {
        double v = G[j][c2];
        for (int k = 0; k < j; ++k) v -= R[k][j] * R[k][c2];
        R[j][c2] = v / rjj;
      }
    }
    // back substitution for R^-1 (columns of dropped vectors stay zero)
    for (int j = 0; j < B; ++j) {
      if (R[j][j] == 0.0) continue;
      Ri[j][j] = 1.0 / R[j][j];
      for (int i = j - 1; i >= 0; --i) {
        if (R[i][i] == 0.0) continue;
        double v = 0.0;
        for (int k = i + 1; k <= j; ++k) v -= R[i][k] * Ri[k][j];
        Ri[i][j] = v / R[i][i];
      }
    }
    flags[0] = mask;
  }
  __syncthreads();
  Rinv[tid] = Ri[tid >> 4][tid & 15];
}

// W <- W * Rinv; optional copies: Qdst[:, col0 + j] and Vs = c .* W
__global__ __launch_bounds__(256) void k_apply_rinv(
    double* __restrict__ W, int n, const double* __restrict__ Rinv,
    double* __restrict__ Qdst, int ldq, int col0, const double* __restrict__ cvec,
    double* __restrict__ Vs) {
  __shared__ double Rl[B * B];
  const int tid = threadIdx.x;
  Rl[tid] = Rinv[tid];
  __syncthreads();
  const int jj = tid & 15;
  const int r = blockIdx.x * 16 + (tid >> 4);
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

// ---------------------------------------------------------------- dense Jacobi
// One workgroup, matrix A (mp x mp, mp = m rounded up to even) in LDS with row
// stride mp + 1; eigenvector accumulator kept TRANSPOSED in global memory
// (Yt[p][i] = component i of vector p) so that the rotation of two vectors is
// two coalesced row updates.  Round-robin ordering: mp/2 disjoint rotations per
// round, mp - 1 rounds per sweep.
// mode 0: A = src (m x m).   mode 1: A_ij = c_i c_j src_ij + delta_ij p_i.
__global__ __launch_bounds__(1024) void k_jacobi(
    const double* __restrict__ src, int ldsrc, int m, int mode,
    const double* __restrict__ cvec, const double* __restrict__ pvec,
    const double* __restrict__ G, double* __restrict__ theta,
    double* __restrict__ Y, int ldy, double* __restrict__ resid,
    double* __restrict__ Yt) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int mp = (m + 1) & ~1;
  const int lda = mp + 1;
  double* A = smem;                       // mp * lda
  double* cs = A + mp * lda;              // mp/2
  double* sn = cs + mp / 2;               // mp/2
  int* pp = reinterpret_cast<int*>(sn + mp / 2);  // mp/2
  int* qq = pp + mp / 2;                  // mp/2
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
    Yt[(size_t)i * mp + j] = (i == j) ? 1.0 : 0.0;
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

  const int half = mp / 2;
  for (int sweep = 0; sweep < 40; ++sweep) {
    if (tid == 0) s_rot = 0;
    __syncthreads();
    for (int round = 0; round < mp - 1; ++round) {
      // --- rotation parameters
      if (tid < half) {
        int p, q;
        if (tid == 0) {
          p = mp - 1;
          q = round;
        } else {
          p = (round + tid) % (mp - 1);
          q = (round - tid + (mp - 1)) % (mp - 1);
        }
        if (p > q) { const int t2 = p; p = q; q = t2; }
        const double app = A[p * lda + p], aqq = A[q * lda + q];
        const double apq = A[p * lda + q];
        double c = 1.0, s = 0.0;
        const double thr = 1e-18 * sqrt(fabs(app) * fabs(aqq));
        if (fabs(apq) > thr && fabs(apq) > 1e-300) {
          const double th = (aqq - app) / (2.0 * apq);
          const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
          if (s != 0.0) atomicAdd(&s_rot, 1);
        }
        cs[tid] = c;
        sn[tid] = s;
        pp[tid] = p;
        qq[tid] = q;
      }
      __syncthreads();
      // --- columns: A <- A J ; vectors: rows p, q of Yt
      for (int e = tid; e < mp * half; e += nth) {
        const int k = e / mp, i = e - k * mp;
        const double c = cs[k], s = sn[k];
        if (s != 0.0) {
          const int p = pp[k], q = qq[k];
          const double aip = A[i * lda + p], aiq = A[i * lda + q];
          A[i * lda + p] = c * aip - s * aiq;
          A[i * lda + q] = s * aip + c * aiq;
          const double yp = Yt[(size_t)p * mp + i], yq = Yt[(size_t)q * mp + i];
          Yt[(size_t)p * mp + i] = c * yp - s * yq;
          Yt[(size_t)q * mp + i] = s * yp + c * yq;
        }
      }
      __syncthreads();
      // --- rows: A <- J^T A
      for (int e = tid; e < mp * half; e += nth) {
        const int k = e / mp, j = e - k * mp;
        const double c = cs[k], s = sn[k];
        if (s != 0.0) {
          const int p = pp[k], q = qq[k];
          const double apj = A[p * lda + j], aqj = A[q * lda + j];
          double npj = c * apj - s * aqj;
          double nqj = s * apj + c * aqj;
          if (j == q) npj = 0.0;
          if (j == p) nqj = 0.0;
          A[p * lda + j] = npj;
          A[q * lda + j] = nqj;
        }
      }
      __syncthreads();
    }
    if (s_rot == 0) break;
    __syncthreads();
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
          const double ya = Yt[(size_t)i * mp + (m - B + a)];
          double t = 0.0;
          for (int b2 = 0; b2 < B; ++b2)
            t = __builtin_fma(G[a * B + b2], Yt[(size_t)i * mp + (m - B + b2)], t);
          r2 = __builtin_fma(ya, t, r2);
        }
      }
      resid[rank] = sqrt(fmax(r2, 0.0));
    }
    // column `rank` of Y = vector i
    for (int r = 0; r < m; ++r) Y[(size_t)r * ldy + rank] = Yt[(size_t)i * mp + r];
  }
}

__global__ void k_set_diag_T(double* T, int ldt, int mtot, const double* theta,
                             int keep) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= mtot * mtot) return;
  const int i = e / mtot, j = e - i * mtot;
  T[(size_t)i * ldt + j] = (i == j && i < keep) ? theta[i] : 0.0;
}

// dst[r, 0:cols] = Q[r, 0:m] * Y[0:m, 0:cols]
__global__ __launch_bounds__(256) void k_basis_times_Y(
    const double* __restrict__ Q, int ldq, int m, const double* __restrict__ Y,
    int ldy, int cols, double* __restrict__ dst, int lddst, int n) {
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
    dst[(size_t)r * lddst + j] = acc;
  }
}

__global__ void k_copy_block(const double* __restrict__ src, int ldsrc,
                             double* __restrict__ dst, int lddst, int n, int cols) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * cols) return;
  const int r = e / cols, j = e - r * cols;
  dst[(size_t)r * lddst + j] = src[(size_t)r * ldsrc + j];
}

// E[:, j] <- t .* E[:, j]; partial column sums of squares per block
__global__ __launch_bounds__(256) void k_scale_colsq(double* __restrict__ E, int lde,
                                                     int n, int cols,
                                                     const double* __restrict__ tvec,
                                                     double* __restrict__ part) {
  __shared__ double sm[256];
  const int tid = threadIdx.x;
  const int rows_per = (n + gridDim.x - 1) / gridDim.x;
  const int rbeg = blockIdx.x * rows_per, rend = min(n, rbeg + rows_per);
  // thread -> (row lane = tid / 64 .. , column = tid % 64)
  const int j = tid & 63, rl = tid >> 6;
  double acc = 0.0;
  if (j < cols) {
    for (int r = rbeg + rl; r < rend; r += 4) {
      const double v = tvec[r] * E[(size_t)r * lde + j];
      E[(size_t)r * lde + j] = v;
      acc = __builtin_fma(v, v, acc);
    }
  }
  sm[tid] = acc;
  __syncthreads();
  if (tid < 64)
    part[(size_t)blockIdx.x * kMaxVectors + tid] =
        (sm[tid] + sm[tid + 64]) + (sm[tid + 128] + sm[tid + 192]);
}
__global__ __launch_bounds__(256) void k_normalize_cols(double* __restrict__ E,
                                                        int lde, int n, int cols,
                                                        const double* __restrict__ part,
                                                        int nparts) {
  __shared__ double inv[kMaxVectors];
  const int tid = threadIdx.x;
  if (tid < cols) {
    double s = 0.0;
    for (int g = 0; g < nparts; ++g) s += part[(size_t)g * kMaxVectors + tid];
    inv[tid] = 1.0 / sqrt(s);
  }
  __syncthreads();
  const size_t total = (size_t)n * cols;
  for (size_t e = (size_t)blockIdx.x * 256 + tid; e < total;
       e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / cols), j = (int)(e - (size_t)r * cols);
    E[(size_t)r * lde + j] *= inv[j];
  }
}

// ---------------------------------------------------------------- launchers
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
  hipLaunchKernelGGL(k_block_matvec, dim3((n + 15) / 16), dim3(256), 0, s, S, ld, n,
                     cvec, pvec, V, ldv, Vs, W);
}
void launch_proj_partial(hipStream_t s, const double* Q, int ldq, int m,
                         const double* W, int n, double* partial) {
  hipLaunchKernelGGL(k_proj_partial, dim3(kProjBlocks), dim3(256), 0, s, Q, ldq, m, W,
                     n, partial);
}
void launch_reduce_H(hipStream_t s, const double* partial, int m, double* Hbuf,
                     double* T, int ldt, int col0, int accumulate, double* hsq) {
  hipLaunchKernelGGL(k_reduce_H, dim3(1), dim3(256), 0, s, partial, kProjBlocks, m,
                     Hbuf, T, ldt, col0, accumulate, hsq);
}
void launch_update_block(hipStream_t s, const double* Q, int ldq, int m,
                         const double* Hbuf, double* W, int n) {
  hipLaunchKernelGGL(k_update_block, dim3((n + 15) / 16), dim3(256), 0, s, Q, ldq, m,
                     Hbuf, W, n);
}
void launch_reduce_chol(hipStream_t s, const double* partial, double* Rinv,
                        double* Gsave, const double* hsq, int* flags) {
  hipLaunchKernelGGL(k_reduce_chol, dim3(1), dim3(256), 0, s, partial, kProjBlocks,
                     Rinv, Gsave, hsq, flags);
}
void launch_apply_rinv(hipStream_t s, double* W, int n, const double* Rinv,
                       double* Qdst, int ldq, int col0, const double* cvec,
                       double* Vs) {
  hipLaunchKernelGGL(k_apply_rinv, dim3((n + 15) / 16), dim3(256), 0, s, W, n, Rinv,
                     Qdst, ldq, col0, cvec, Vs);
}
void launch_jacobi(hipStream_t s, const double* src, int ldsrc, int m, int mode,
                   const double* cvec, const double* pvec, const double* G,
                   double* theta, double* Y, int ldy, double* resid, double* Yt) {
  const int mp = (m + 1) & ~1;
  const size_t lds = sizeof(double) * ((size_t)mp * (mp + 1) + mp) +
                     sizeof(int) * (size_t)mp + 64;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_jacobi),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_jacobi, dim3(1), dim3(1024), lds, s, src, ldsrc, m, mode, cvec,
                     pvec, G, theta, Y, ldy, resid, Yt);
}
void launch_set_diag_T(hipStream_t s, double* T, int ldt, int mtot,
                       const double* theta, int keep) {
  hipLaunchKernelGGL(k_set_diag_T, dim3((mtot * mtot + 255) / 256), dim3(256), 0, s, T,
                     ldt, mtot, theta, keep);
}
void launch_basis_times_Y(hipStream_t s, const double* Q, int ldq, int m,
                          const double* Y, int ldy, int cols, double* dst,
                          int lddst, int n) {
  const size_t lds = sizeof(double) * ((size_t)m * cols + 16 * (size_t)(m + 1));
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_basis_times_Y),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_basis_times_Y, dim3((n + 15) / 16), dim3(256), lds, s, Q, ldq,
                     m, Y, ldy, cols, dst, lddst, n);
}
void launch_copy_block(hipStream_t s, const double* src, int ldsrc, double* dst,
                       int lddst, int n, int cols) {
  hipLaunchKernelGGL(k_copy_block, dim3((n * cols + 255) / 256), dim3(256), 0, s, src,
                     ldsrc, dst, lddst, n, cols);
}
void launch_back_transform(hipStream_t s, double* E, int lde, int n, int cols,
                           const double* tvec, double* colnorm_ws) {
  hipLaunchKernelGGL(k_scale_colsq, dim3(kProjBlocks), dim3(256), 0, s, E, lde, n,
                     cols, tvec, colnorm_ws);
  hipLaunchKernelGGL(k_normalize_cols, dim3(256), dim3(256), 0, s, E, lde, n, cols,
                     colnorm_ws, kProjBlocks);
}

}  // namespace sc
