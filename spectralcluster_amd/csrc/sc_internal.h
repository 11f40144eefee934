// Internal declarations shared by the HIP translation units of
// libspectralcluster_amd.so.  gfx950 (MI355X) only: wave64, fp64 MFMA.
#pragma once

#include <hip/hip_runtime.h>

#include "switches.h"
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/spectralcluster_amd.h"

namespace sc {

constexpr int kWave = 64;
constexpr int kEigBlock = 8;         // Lanczos block width (half an MFMA tile column)
constexpr int kEigBasisCap = 128;    // Rayleigh-Ritz size limit (LDS Jacobi)
constexpr int kLdq = kEigBasisCap + kEigBlock;  // row stride of the Krylov basis
constexpr int kDenseMax = 128;       // n <= this: direct dense Jacobi
constexpr int kHostRR = 64;          // lockstep group solve: Rayleigh-Ritz problems up to this
                                     // order are solved on the host (O(m^3) scalars, like the
                                     // eigengap loop); larger ones leave the group
constexpr int kHostRRSingle = kEigBasisCap;  // single-call solve: every check on the host (the
                                     // one-workgroup Jacobi takes 1.7 / 2.7 / 4.7 / 6.4 ms at
                                     // m = 80 / 96 / 112 / 128; tred2 + tql2 on a host core a
                                     // fraction of that) -- clustered spectra reach these sizes
constexpr int kGenMax = 64;          // general eigen path: dense limit and Arnoldi basis cap
constexpr int kMaxVectors = 64;      // eigenvector columns kept resident
constexpr int kProjBlocks = 128;     // partial-sum blocks for tall-skinny products

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device, per-function attribute: set it
// once per device (thread-safe; handles of several devices / threads share the process)
#define SC_OPT_IN_LDS(kernel, bytes)                                                      \
  do {                                                                                    \
    static std::once_flag sc_once_[16];                                                   \
    int sc_dev_ = 0;                                                                      \
    (void)hipGetDevice(&sc_dev_);                                                         \
    std::call_once(sc_once_[sc_dev_ & 15], [] {                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    });                                                                                   \
  } while (0)

// GEMM epilogues
enum { kEpiNone = 0, kEpiAffinity = 1, kEpiAdd = 2 };

// ---- kernel launchers (each enqueues on `s`, no sync) -----------------------
// C[M,N] = A[M,K] * B[N,K]^T (row-major, leading dims in elements).  When
// `symmetric` is set the caller asserts that the product is symmetric (B aliasing A, or
// commuting symmetric operands): only tile pairs i<=j are computed and the mirror tile
// is written transposed, so C is exactly symmetric.  kEpiAdd adds `addend` (ld = ldc,
// symmetric when `symmetric`; must not alias C).
// `splitk_ws`: gemm_splitk_workspace_bytes() of scratch owned by the caller (per handle).
// `rs` (optional): row statistics of C fused into the epilogue -- mode 1: rowmax / rowsum,
// mode 2: max over j != i clamped at 0 (CropDiagonal's fill value) in rowmax.  The partial
// arrays hold M x gemm_tile_dim(N) doubles each.
struct GemmRowStats {
  int mode;
  double* partial_max;
  double* partial_sum;
  double* rowmax;
  double* rowsum;
};
void launch_gemm_nt(hipStream_t s, const double* A, int lda, const double* B,
                    int ldb, double* C, int ldc, int M, int N, int K,
                    int epilogue, bool symmetric, double* splitk_ws,
                    const int2* tilemap, const GemmRowStats* rs = nullptr,
                    const double* addend = nullptr);
// One member of a grouped symmetric product C = A A^T (launch_gemm_nt_group); n = 0: idle.
// partial_max / partial_sum: n x gemm_tile_dim(n) doubles each (sum only for stats mode 1).
struct GemmGroupItem {
  const double* A = nullptr;
  int lda = 0;
  double* C = nullptr;
  int ldc = 0;
  int n = 0, K = 0;
  const int2* tilemap = nullptr;
  double* partial_max = nullptr;
  double* partial_sum = nullptr;
  double* rowmax = nullptr;
  double* rowsum = nullptr;
};
void launch_gemm_nt_group(hipStream_t s, const GemmGroupItem* items, int count, int epilogue,
                          int stats_mode);
// patch-ordered (ti, tj) list of the upper triangle, for `tilemap` (symmetric launches)
void gemm_build_sym_tilemap(int nt, std::vector<int2>* out);
int gemm_tile_dim(int n);
int gemm_resident_slots();
size_t gemm_splitk_workspace_bytes();

// bad_rows (optional): set to 1 when a row has zero or non-finite norm (its affinities are NaN)
void launch_normalize_rows(hipStream_t s, const double* X, int ldx, int n, int d,
                           double* Xn, int* bad_rows = nullptr);
void launch_crop_diagonal(hipStream_t s, const double* in, double* out, int n,
                          int ld);
void launch_gaussian_blur(hipStream_t s, const double* in, double* out, int n,
                          int ld, int radius, const double* weights_dev);
// fused forms used by the predict() pipeline
bool launch_gaussian_blur_fused(hipStream_t s, const double* in, double* out, int n, int ld,
                                int radius, const double* weights_dev, const double* diag,
                                double* rowmax_partials);
int blur_tile_columns(int n, int radius);
// radius above SC_MAX_BLUR_RADIUS (sigma > 8): two global passes, tmp = an n x ld scratch
void launch_gaussian_blur_any_radius(hipStream_t s, const double* in, double* tmp, double* out,
                                     int n, int ld, int radius, const double* weights_dev);
void launch_crop_value(hipStream_t s, const double* in, int n, int ld, double* dvec);
void launch_cut_from_partials(hipStream_t s, const double* partials, int n, int ntiles,
                              double p, double* cut);
void launch_cut_from_rows(hipStream_t s, const double* in, int n, int ld, double p,
                          double* cut, int zero_diag);
void launch_cut_percentile(hipStream_t s, const double* in, int n, int ld, double p,
                           double* cut, int zero_diag);
void launch_row_threshold_cut(hipStream_t s, const double* in, double* out, int n, int ld,
                              const double* cut, double mult, int binarize,
                              int preserve_diag);
void launch_threshold_symmetrize(hipStream_t s, const double* in, double* out, int n, int ld,
                                 const double* cut, double mult, int binarize, int symtype,
                                 int preserve_diag);
// ... + the digits and row partials of the matrix-free Diffuse's quantiser (rowops.hip)
void launch_threshold_symmetrize_digits(hipStream_t s, const double* in, double* out, int n, int ld,
                                        const double* cut, double mult, int binarize, int symtype,
                                        int preserve_diag, signed char* Q, double* scal,
                                        double* ypart, int* rpart, double* q2part,
                                        double* mx64);
struct FreeSegs;  // (below, with the matrix-free Diffuse)
void launch_free_partials_reduce(hipStream_t s, const double* ypart, const int* rpart, int n,
                                 double* y1, double* R, double* scal, const FreeSegs& segs);
void launch_row_threshold(hipStream_t s, const double* in, double* out, int n,
                          int ld, double p, double mult, int binarize,
                          int preserve_diag);
void launch_symmetrize(hipStream_t s, const double* in, double* out, int n, int ld,
                       int type);

// constraint.hip (reference constraint.py:95-164)
void launch_affinity_integration(hipStream_t s, const double* a, const double* q, double* out,
                                 int n, int ld, int type);
void launch_cp_prepare(hipStream_t s, const double* a, const double* deg, double alpha,
                       double* p, double* t0, int n, int ld);
void launch_cp_adjust(hipStream_t s, const double* tqt, const double* a, double scale,
                      double* out, int n, int ld);
void launch_transpose(hipStream_t s, const double* in, double* out, int n, int ld);
void launch_symmetry_flag(hipStream_t s, const double* in, int n, int ld, int* flag);
void launch_row_normalize(hipStream_t s, const double* in, double* out, int n,
                          int ld);
void launch_row_stats(hipStream_t s, const double* in, int n, int ld,
                      double* rowmax, double* rowsum);
// c/p/t vectors of the symmetric operator  Op = diag(p) + diag(c) S diag(c)
void launch_scaling_vectors(hipStream_t s, const double* rowmax,
                            const double* rowsum, int n, int laplacian_type,
                            int row_normalized, double* c, double* p, double* t,
                            int* flags = nullptr, int* symflag = nullptr);
void launch_laplacian(hipStream_t s, const double* in, double* out, int n, int ld,
                      int laplacian_type, double* deg_ws);
// *flag = 1 if a or b holds a NaN / inf
void launch_check_finite(hipStream_t s, const double* a, const double* b, int n, int* flag);

// ---- eigensolver -------------------------------------------------------------
struct EigWorkspace {
  double* Q = nullptr;        // n x kLdq Krylov basis (row-major)
  double* Q2 = nullptr;       // restart scratch, same shape
  double* Vs = nullptr;       // n x 16: c .* current block
  double* W = nullptr;        // n x 16: operator output / next block
  double* partial = nullptr;  // kProjBlocks x (kLdq*16)
  double* T = nullptr;        // kLdq x kLdq projected operator
  double* Y = nullptr;        // kLdq x kLdq Ritz coefficient vectors
  double* theta = nullptr;    // kLdq Ritz values (descending)
  double* resid = nullptr;    // kLdq residual estimates
  double* G = nullptr;        // 16x16 Gram of the residual block
  double* Rinv = nullptr;     // 16x16
  double* Hbuf = nullptr;     // kLdq x 16 projection coefficients of one pass
  double* hsq = nullptr;      // 16 column energies removed by projection
  double* Yt = nullptr;       // kLdq x kLdq Jacobi vector accumulator (transposed)
  double* colnorm = nullptr;  // kProjBlocks x kMaxVectors partial column sums
  int* flags = nullptr;       // [0] rank-deficiency mask of the last block
};

void launch_random_block(hipStream_t s, double* W, int n, uint64_t seed);
void launch_block_matvec(hipStream_t s, const double* S, int ld, int n,
                         const double* cvec, const double* pvec, const double* V,
                         int ldv, const double* Vs, double* W);
// the same from the upper triangle of a symmetric S (half the HBM bytes); ws:
// matvec_sym_workspace_doubles(n) doubles of slabs
size_t matvec_sym_workspace_doubles(int n);
void launch_block_matvec_sym(hipStream_t s, const double* S, int ld, int n, const double* cvec,
                             const double* pvec, const double* V, int ldv, const double* Vs,
                             double* W, double* ws);
void launch_proj_partial(hipStream_t s, const double* Q, int ldq, int m,
                         const double* W, int n, double* partial);
// H = sum of partials (m x 16) -> Hbuf; if T != nullptr also (accumulated) into
// T[0:m, col0:col0+16] and mirrored; hsq[j] (+)= sum_i H_ij^2.
int proj_blocks(int n);
void launch_reduce_H(hipStream_t s, const double* partial, int nparts, int m, double* Hbuf,
                     double* T, int ldt, int col0, int accumulate, double* hsq);
void launch_update_block(hipStream_t s, const double* Q, int ldq, int m,
                         const double* Hbuf, double* W, int n);
// Gram reduce + Cholesky: Rinv (16x16 upper), optional copy of G, flags mask.
// defect_flag (optional): set to 1 when the block was ill-conditioned (pivot ratio > 1e3)
// or, with flag_mode 2, when the input Gram matrix is far from I (see k_reduce_chol)
void launch_reduce_chol(hipStream_t s, const double* partial, int nparts, double* Rinv,
                        double* Gsave, const double* hsq, int* flags,
                        int* defect_flag = nullptr, int flag_mode = 1);
// W <- W * Rinv ; optionally also store into Q[:, col0:col0+16] and Vs = c .* W
void launch_apply_rinv(hipStream_t s, double* W, int n, const double* Rinv,
                       double* Qdst, int ldq, int col0, const double* cvec,
                       double* Vs);
void launch_refill_deficient(hipStream_t s, double* W, int n, const int* flags,
                             uint64_t seed);
// Orthonormalisation chain of short kernels (k_lz_rows in eig.hip): no host sync.
// ws.partial must hold lz_partial_doubles(n) doubles (two buffers, used alternately).
struct LzChain {
  int parity = 0;   // partial buffer the last link wrote
  int nparts = 0;   // workgroups of the last link
  // CholQR3 form of a block step: links 0>1, 1>2, 2>3, 3>3, 3>0(store) instead of 0>1, 1>2,
  // 2>3, 3>0(store).  A Krylov block of condition > ~1e8 (a numerically low-rank operator:
  // few dominant eigenvalues, the rest 1e-8 below) leaves the first Cholesky pass with an
  // orthogonality error that two passes do not remove; three do.
  bool three_pass = false;
};
size_t lz_partial_doubles(int n);
void launch_lz_link(hipStream_t s, const EigWorkspace& ws, LzChain* chain, int n, int m,
                    int pre, int next, int store_col, const double* vs_scale, int col0,
                    bool init_random, uint64_t seed, bool zero_T);
// ---- grouped launches (batch_group.hip): up to kGroupMax independent problems per launch,
//      their descriptors passed by value in the kernel arguments (<= 4 KB), blockIdx.y = member
constexpr int kGroupMax = 16;
template <typename T>
struct GroupOf {
  T s[kGroupMax];
};
struct MatvecItem {  // n = 0: idle member
  const double* S;
  int ld, n;
  const double* cvec;
  const double* pvec;
  const double* V;
  int ldv;
  const double* Vs;
  double* W;
  double* slabs;  // matvec_sym_workspace_doubles(n) doubles (symmetric form only)
};
struct RitzItem {    // E[:, 0:cols] (column-major, lde) = normalise(t .* (Q[:, 0:m] Y[:, 0:cols]))
  const double* Q;
  int ldq, m;
  const double* Y;
  int ldy, cols;
  double* E;
  int lde, n;
  const double* tvec;
};
// One member of a batch group in the stages before its eigensolver (ICASSP2018 refinement
// sequence with the fused kernels): where its arena keeps what.  n = 0: idle.
struct FrontItem {
  const double* X;   // (n, d) embeddings, row pitch ldx
  double* Xn;        // unit rows
  int ldx, n, d, ldn;
  double* A0;        // affinity
  double* B1;        // blurred, later the Diffuse result
  double* B2;        // thresholded + symmetrised
  const double* cropval;
  double* rmpart;    // per-strip row maxima of the blur
  int blur_cols;     // strips per row (blur_tile_columns)
  double* cut;
  const double* rowmax;
  const double* rowsum;
  double* cvec;
  double* pvec;
  double* tvec;
  int* symflag;      // [1]: a zero / non-finite embedding row was seen (may be null)
  int* flags;
  double p_own;      // > 0: this member's own p_percentile (AutoTune sweep)
};
void launch_front_begin_group(hipStream_t s, const FrontItem* items, int count,
                              bool normalize_rows);
void launch_cut_percentile_group(hipStream_t s, const FrontItem* items, int count, int zero_diag);
void launch_row_stats_group(hipStream_t s, const FrontItem* items, int count);
bool blur_group_supported(int n_min, int radius);
// grouped front of a batch (16 members per launch): from n = 256 on; its strips per row
bool blur_group_front_supported(int n_min, int radius);
int blur_stream_columns(int n, int radius);
void launch_gaussian_blur_group(hipStream_t s, const FrontItem* items, int count, int radius,
                                const double* weights_dev);
// What the threshold + symmetrise pass leaves for the matrix-free Diffuse when it writes the digits
// itself (rowops.hip threshold_symmetrize_body<true>; Q == nullptr: this matrix gets none): the
// two 8-bit digits of q = rint(sigma a) in the product's layout (a 64 x 64 tile is one 128-byte
// line per row: 64 high digits, 64 low digits) and, per row and 64-column block, the partial
// sums of a, of |q| and of q^2 (k_free_partials_reduce adds them in block order), and per
// 64 x 64 tile its largest segment norm.  Saves one read of the matrix.
struct TsDigits {
  signed char* Q;      // digits, row pitch `pitch` bytes
  size_t pitch;
  int nblk;            // 64-column blocks per row
  double* scal;        // [0] = max |a| (known from the cut vector before this pass); [3] set when
                       // a finite value had to be clamped after all (diffuse_free.hip)
  double* ypart;       // [block * 64 nblk + row] sum of a  (a tile's 64 partials are contiguous:
  int* rpart;          // [block * 64 nblk + row] sum of |q|   whole lines leave the L2)
  double* q2part;      // [block * 64 nblk + row] sum of q^2 (an exact integer < 2^37): the squared
                       // norm of the row's digit segment, for the tile skip list (diffuse_free.hip)
  double* mx64;        // [row group * nblk + block] the largest segment norm of a tile's 64 rows
                       // (rounded up): exactly one workgroup holds the 64 values, a plain store
};
// (digits: per member what the pass writes besides the matrix, or nullptr -- no member gets any)
void launch_threshold_symmetrize_group(hipStream_t s, const FrontItem* items, int count,
                                       double p, double mult, int binarize, int symtype,
                                       int preserve_diag, bool cut_ready = false,
                                       const TsDigits* digits = nullptr);
void launch_cut_from_partials_group(hipStream_t s, const FrontItem* items, int count, double p);
void launch_scaling_vectors_group(hipStream_t s, const FrontItem* items, int count,
                                  int laplacian_type, int row_normalized);
struct GatherItem {  // T == nullptr: idle
  const double* T;
  const double* G;
  const int* flags;
};
void launch_group_gather(hipStream_t s, const GatherItem* items, int count, int m, double* out,
                         int stride);
struct LzGroupMember {
  EigWorkspace ws;
  LzChain chain;
  int n = 0;
  const double* vs_scale = nullptr;
  bool active = false;
};
void launch_lz_link_group(hipStream_t s, LzGroupMember* mem, int count, int m, int pre,
                          int next, int store_col, int col0, bool init_random, uint64_t seed,
                          bool zero_T);
void launch_block_matvec_group(hipStream_t s, const MatvecItem* items, int count,
                               bool symmetric);
void launch_ritz_vectors_group(hipStream_t s, const RitzItem* items, int count);
// Dense symmetric eigensolver (one workgroup, cyclic Jacobi, matrix in LDS).
// mode 0: A = T (m x m, ldt).  mode 1: A_ij = c_i c_j S_ij + delta_ij p_i.
void launch_jacobi(hipStream_t s, const double* src, int ldsrc, int m, int mode,
                   const double* cvec, const double* pvec, const double* G,
                   double* theta, double* Y, int ldy, double* resid, double* Yt,
                   int* dbg);
void launch_set_diag_T(hipStream_t s, double* T, int ldt, int mtot,
                       const double* theta, int keep);
// dst[:, 0:cols] = Q[:, 0:m] * Y[0:m, 0:cols]
void launch_basis_times_Y(hipStream_t s, const double* Q, int ldq, int m,
                          const double* Y, int ldy, int cols, double* dst,
                          int lddst, int n, int colmajor);
void launch_copy_block(hipStream_t s, const double* src, int ldsrc, double* dst,
                       int lddst, int n, int cols);
// Eigenvectors live COLUMN-major on the device: ET[j * ld + r].
// ET[j] = t .* ET[j] / || t .* ET[j] ||   (LAPACK unit 2-norm columns)
void launch_back_transform(hipStream_t s, double* ET, int ld, int n, int cols,
                           const double* tvec);
void launch_rowmajor_to_colmajor(hipStream_t s, const double* src, int lds, int n, int cols,
                                 double* dst, int ldd);
void launch_colmajor_to_rowmajor(hipStream_t s, const double* src, int lds, int n, int cols,
                                 double* dst, int ldd);

// ---- dense full-spectrum symmetric path, eig_dense.hip ------------------------------
// M = diag(p) + diag(c) S diag(c), exactly symmetric
void launch_td_materialize(hipStream_t s, const double* S, int ld, int n, const double* c,
                           const double* p, double* M);
// Householder tridiagonalisation of A (n x n, ld; destroyed) in panels of 32 columns (LAPACK
// dsytrd / dlatrd): d[0..n), e[0..n-1), taus[0..n); reflector j (v[j+1] = 1) is left in
// A[j, j+1 .. n-1].  8 B of read-only traffic per trailing entry and column, the trailing
// update once per panel on the MFMA GEMM.  panel: 4 * 32 * n doubles; work: 5 n + 2048.
void launch_tridiagonalize_blocked(hipStream_t s, double* A, int ld, int n, double* d, double* e,
                                   double* taus, double* panel, double* work,
                                   double* splitk_ws);
// Z (column-major: column q at Z + q * ldz) <- Q Z, Q = H_0 ... H_{n-2} from the above
void launch_td_backtransform(hipStream_t s, const double* A, int ld, int n, const double* taus,
                             double* Z, int ldz, int cols);
// all eigenvalues of the tridiagonal (d, e) by Sturm bisection, DESCENDING; work: n + 4
void launch_tridiagonal_eigenvalues(hipStream_t s, const double* d, const double* e, int n,
                                    double* theta_desc, double* work);

// ---- general (non-symmetric) eigen path, eig_general.hip ---------------------------
// Dense complex-Schur eigensolver, one wavefront, order m <= kGenMax: eigenvalues of
// sign * A sorted by real part (descending) into theta_re / theta_im, the first `nvec`
// unit-norm eigenvectors into Y[:, q] = Yre + i Yim (row-major, ldy).  info[0] != 0: the QR
// iteration failed; info[1] = sweeps.
void launch_gen_eig(hipStream_t s, const double* A, int lda, int m, double sign, int nvec,
                    double* theta_re, double* theta_im, double* Yre, double* Yim, int ldy,
                    int* info);
int gen_residual_blocks(int n);
// resid[c] = || OpQ y_c - theta_c Q y_c ||_2 for c < cols (<= 32); partial: blocks x 32
void launch_gen_residual(hipStream_t s, const double* Q, const double* OpQ, int ldq, int m,
                         int n, const double* Yre, const double* Yim, int ldy,
                         const double* theta_re, const double* theta_im, int cols,
                         double* partial, double* resid);
// V[:, c] = Q y_c (complex, column-major ldv); Q == nullptr: V = Y
void launch_gen_ritz(hipStream_t s, const double* Q, int ldq, int m, int n, const double* Yre,
                     const double* Yim, int ldy, int cols, double* Vre, double* Vim, int ldv);
// dgeev's normalisation per column (unit norm, largest component real); E (may be null)
// receives the real parts, column-major lde
void launch_gen_phase(hipStream_t s, double* Vre, double* Vim, int ldv, int n, int cols,
                      double* E, int lde);
// hessenberg.hip: B (n x n, ld) -> upper Hessenberg form in place, reflector k below the
// subdiagonal of column k (LAPACK dgehd2's storage), tau (n), vwork (2 n doubles)
void launch_hessenberg(hipStream_t s, double* B, int ld, int n, double* tau, double* vwork);
void launch_gen_gather(hipStream_t s, const double* Vre, const double* Vim, int ldv, int n,
                       const int* src, uint64_t seed, double* W);
void launch_scale_matrix(hipStream_t s, double* A, int ld, int n, double f);
// out[0 .. n) = -a, out[stride .. stride + n) = -b  (the operator of gen_topk with its sign turned)
void launch_negate2(hipStream_t s, const double* a, const double* b, int n, double* out,
                    size_t stride);
void launch_scaling_general(hipStream_t s, const double* deg, int n, int laplacian_type,
                            double* cl, double* cr, double* p);

// ---- matrix-free Diffuse (diffuse_free.hip; host side free_api.hip) ---------------------
// rowmax / rowsum of S = A A^T without the fp64 product: 8-bit fixed-point digits of A, an exact
// integer MFMA product of them, row maxima + candidates within a proven slack, exact fp64 dot
// products for the candidates.  scal: 4 doubles ([0] max|a| bits, [2] max R bits; zeroed by
// the caller); M / count: n words each, zeroed; ovf: 80 words, zeroed; cand: n x cap.
// The candidate threshold of a row whose maximum of T is m (diffuse_free.hip's header: twice the
// bound on |sigma^2 S - T| with R_j replaced by its maximum, plus the fp32 rounding of the two
// stored values that are compared -- 2^-24 relative each, doubled for safety).  Monotone in m,
// decreasing in Rmax; rounded DOWN to fp32 (the comparison is made in fp32).
__device__ __forceinline__ float free_threshold(float m, double Ri, double Rmax, int n) {
  const double e = 0.5000001 * (Ri + Rmax) + 0.26 * (double)n;
  const double thr = (double)m - 2.0 * e - 2.4e-7 * fabs((double)m);
  float t = (float)thr;
  if ((double)t > thr) t = nextafterf(t, -INFINITY);
  return t;
}
// Tile skip list of the digit product (diffuse_free.hip, "tile pruning"): what the pass that
// produces the digits leaves per 64-row group g and 64-column block b
//   mx64[g * nblk + b] = max over the group's rows of ||q_i[block b]||_2      (0 for rows >= n)
//   tau64[g]           = min over the group's rows < n of the candidate threshold the row would
//                        have if its maximum of T were only its diagonal entry T_ii = ||q_i||^2
//                        and R_max the trivial bound 32639 n  (+inf for a group without rows)
struct FreeSegs {
  const double* q2part;  // [block * 64 nblk + row]: sum of q^2 over the block (exact integers)
  double* mx64;
  float* tau64;
};
int free_rows_padded(int n);
int free_k_padded(int n);
size_t free_q_bytes(int n);
size_t free_t32_bytes(int n);
int free_candidate_cap();
void launch_free_absmax(hipStream_t s, const double* A, int n, int ld, double* scal);
// scal[0] = an attained upper bound of max|a| for A = Symmetrize(RowWiseThreshold(B)), B >= 0,
// from the threshold stage's cut vector (cut_i = rowmax(B)_i * p): no pass over the matrix
void launch_free_amax_from_cut(hipStream_t s, const double* cut, int n, double p,
                               double floor_value, double* scal);
// (q2part: [block][row] squared digit-segment norms for the tile skip list, or nullptr)
void launch_free_quantize(hipStream_t s, const double* A, int n, int ld, signed char* Q,
                          double* scal, double* y1, double* R, double* q2part = nullptr);
// tile skip list of the product (diffuse_free.hip "tile skip list"): mx64 / tau64 from the pass
// that wrote the digits (FreeSegs), then plan = per tile row the surviving tile columns
size_t free_q2part_bytes(int n);
size_t free_mx64_bytes(int n);
size_t free_tau64_bytes(int n);
size_t free_plan_bytes(int n);
size_t free_i8_split_bytes_plan();
void launch_free_seg_reduce(hipStream_t s, const double* R, int n, const FreeSegs& segs);
// (words: the handle's M | count | ovf words, zeroed for this call: ovf[68] receives the length)
void launch_free_tile_flags(hipStream_t s, const double* mx64, const float* tau64, int n,
                            int* plan, bool prune, int* words);
// (split_ws: workspace of free_i8_split_bytes(n) for the split-K tail; nullptr = every tile
//  by one workgroup.  plan: walk the skip list instead of `tilemap` -- split_ws must then hold
//  free_i8_split_bytes_plan())
void launch_gemm_i8_sym(hipStream_t s, const signed char* Q, int n, const int2* tilemap,
                        float* T32, unsigned* M, int* split_ws, const int* plan = nullptr);
// one member of a grouped run of the pipeline (AutoTune sweep, large members of a batch group)
struct FreeItem {
  const double* A;   // the symmetric matrix, row pitch ld
  int n, ld;         // n = 0: idle member
  signed char* Q;
  float* T32;
  int* words;        // M (n) | candidate counts (n) | overflow record (80)
  double* scal;      // [0] max|a|, [2] max R
  double* y1;
  double* R;
  int* cand;
  double* rowmax;
  double* rowsum;
  const double* cut; // RowMax cut vector (max|a| = max cut / p) and its p
  double p;
  // tile skip list of the member's digit product (nullptr: every tile)
  double* q2part = nullptr;
  double* mx64 = nullptr;
  float* tau64 = nullptr;
  int* plan = nullptr;
  // row partials when the grouped threshold pass wrote the member's digits (TsDigits)
  double* ypart = nullptr;
  int* rpart = nullptr;
};
// y1, R, max R and the skip list's thresholds of every member from the partials of a grouped
// threshold pass that wrote the digits (the group form of launch_free_partials_reduce)
void launch_free_partials_reduce_group(hipStream_t s, const FreeItem* items, int count);
// (seg_reduce: the members' digits came from the quantiser, whose segment maxima are not formed
//  by the pass itself)
void launch_free_tile_flags_group(hipStream_t s, const FreeItem* items, int count, bool prune,
                                  bool seg_reduce);
void launch_free_begin_group(hipStream_t s, const FreeItem* items, int count, double floor_value,
                             bool pad_rows = false);
void launch_free_quantize_group(hipStream_t s, const FreeItem* items, int count);
void launch_free_scan_stats_group(hipStream_t s, const FreeItem* items, int count);
void free_i8_split_plan(int n, int* tail_tiles, int* parts);
size_t free_i8_split_bytes(int n);
// the same product for `count` (<= kGroupMax) problems of one size in ONE launch
// (plans: per member its skip list, or nullptr -- the array or an entry -- for every tile)
void launch_gemm_i8_sym_group(hipStream_t s, const signed char* const* Q, float* const* T32,
                              unsigned* const* M, int count, const int* ns,
                              const int2* const* tilemaps, const int* const* plans = nullptr);
void launch_t32_candidates(hipStream_t s, const float* T32, int n, const unsigned* M,
                           const double* R, const double* scal, int* count, int* cand,
                           const int* plan = nullptr);
void launch_free_row_stats(hipStream_t s, const double* A, int n, int ld, const double* y1,
                           const int* count, const int* cand, double* rowmax, double* rowsum,
                           int* ovf);
void launch_free_gather_rows(hipStream_t s, const double* A, int n, int ld, const int* rows,
                             int nrows, double* Vs);
void launch_free_colmax(hipStream_t s, const double* W, int n, const int* rows, int nrows,
                        double* rowmax);

// ---- size reduction (ahc.hip) ----------------------------------------------------------
void launch_cosine_distance(hipStream_t s, double* c, int n, int ld);
// nearest-neighbour-chain agglomeration on the n x n distance matrix D (destroyed);
// method 1 complete, 2 average; Z: (n - 1) x 4 (slot x, slot y, height, size), merge order
void launch_ahc_nn_chain(hipStream_t s, double* D, int ld, int n, int method, int* size,
                         int* chain, double* Z);
void launch_cluster_centroids(hipStream_t s, const double* X, int ldx, int n, int d,
                              const int* labels, int k, double* out);

// ---- fallback decisions (fallback.hip) ---------------------------------------------------
void launch_naive_cluster(hipStream_t s, const double* X, int n, int d, double threshold,
                          double adapt_threshold, double* centroids, int* counts,
                          int* n_centroids, int* labels);
// out = {min, min of the first superdiagonal, mean, population std}; partial: n x 8 doubles
void launch_affinity_stats(hipStream_t s, const double* a, int n, int ld, double* partial,
                           double* out);
// one pass of the 1-D mixture over a[i][j], j >= i + offset; sums: 7 doubles (see fallback.hip)
void launch_gmm_pass(hipStream_t s, const double* a, int n, int ld, int offset, int components,
                     int mode, const double* params, double* partial, double* sums);
void launch_gmm_range(hipStream_t s, const double* a, int n, int ld, int offset,
                      double* partial, double* out);

// ---- k-means -------------------------------------------------------------------
struct KmeansWorkspace {
  double* Xc = nullptr;       // kMaxVectors x n centred copy (column-major)
  double* xsq = nullptr;      // n
  double* closest = nullptr;  // n
  double* cand = nullptr;     // 8 x n candidate distances
  double* enorm = nullptr;    // n row norms of E
  double* rnd = nullptr;      // MT19937 doubles (device copy)
  double* centroids = nullptr;  // kMaxVectors x kMaxVectors
  int* labels32 = nullptr;    // n
  long long* labels64 = nullptr;  // n
  int* info = nullptr;        // [0] iterations, [8] chain: stop rule met
  double* chain = nullptr;    // kmeans_chain_workspace_doubles(n): partials of the kernel chain
  double* big = nullptr;      // k > kMaxVectors: kmeans_big_workspace_doubles(k) (else unused)
  int* big_words = nullptr;   // ... and 3 k ints
};
size_t kmeans_big_workspace_doubles(int k);
// k-means as a chain of short multi-workgroup kernels (kmeans_chain.hip), cosine metric
size_t kmeans_chain_workspace_doubles(int n);
bool kmeans_chain_supported(int n, int k, int trials);
struct KmGroupItem {  // one member of a grouped chain launch; n = 0: idle
  const double* ET = nullptr;
  int lde = 0, n = 0, k = 0, max_iter = 0, first_center = 0, trials = 0;
  KmeansWorkspace ws;
};
void launch_kmeans_chain_group(hipStream_t s, const KmGroupItem* items, int count, int it_begin,
                               int it_count);
void launch_kmeans_chain(hipStream_t s, const double* ET, int lde, int n, int k, int max_iter,
                         int first_center, int trials, const KmeansWorkspace& ws, int it_begin,
                         int it_count);
void launch_row_renorm(hipStream_t s, double* ET, int lde, int n, int k);
void launch_to_colmajor(hipStream_t s, const double* src, int n, int k, double* dst,
                        int ldt);
// metric of the custom-distance loop (scipy cdist names)
enum {
  kKmeansCosine = 0,
  kKmeansEuclidean = 1,
  kKmeansSqeuclidean = 2,
  kKmeansCityblock = 3,
  kKmeansChebyshev = 4,
  kKmeansCorrelation = 5,
  kKmeansBraycurtis = 6,
  kKmeansCanberra = 7
};
void launch_kmeans(hipStream_t s, const double* ET, int lde, int n, int k,
                   int max_iter, int first_center, int trials,
                   const KmeansWorkspace& ws, int metric = kKmeansCosine);

}  // namespace sc
