// Host-side dense kernels of the eigen drivers (host_eig.cpp): the O(m^3) scalar solves of the
// projected (Rayleigh-Ritz) problems, m <= 128, and the O(n) inverse iteration on a tridiagonal
// form.  Plain C++ (no device code), compiled with the host compiler.
#ifndef SPECTRALCLUSTER_AMD_HOST_EIG_H_
#define SPECTRALCLUSTER_AMD_HOST_EIG_H_

#include <cstddef>
#include <vector>

// a: m x m symmetric (row-major, lda), overwritten with the eigenvectors (columns);
// d: eigenvalues ascending; e: work (m).  false: QL did not converge.
bool host_symmetric_eig(double* a, int lda, int m, double* d, double* e);

// eigenvectors of the symmetric tridiagonal (d, e) for the k eigenvalues lam (inverse
// iteration, LAPACK dstein's method); Z column-major: column q at Z + q * ldz
bool host_tridiag_eigvectors(const double* d, const double* e, int n, const double* lam, int k,
                             double* Z, size_t ldz);

struct HostTridiag {
  int m = 0, lda = 0;
  std::vector<double> a, v, d, e, tau, theta;  // v: reflector i in row i, columns i+1 ..
};

// Step 1: every eigenvalue of the m x m symmetric T (upper triangle given, row-major ld),
// DESCENDING into w->theta.  Step 2: eigenvectors of the leading `need` eigenvalues into Y
// (row-major ldy: column q).
bool host_partial_values(const double* T, int ld, int m, HostTridiag* w);
bool host_partial_vectors(const HostTridiag& w, int need, double* Y, int ldy);

// General (non-symmetric) real a (m x m, row-major lda): eigenvalues sorted by real part,
// descending, into wr / wi; the first nvec eigenvectors (unit 2-norm) into Y[:, q] = yre + i yim
// (row-major, ldy).  Hessenberg + shifted complex QR + back substitution.  false: no convergence.
bool host_general_eig(const double* a, int lda, int m, int nvec, double* wr, double* wi,
                      double* yre, double* yim, int ldy);

// The same contract by real arithmetic: Householder Hessenberg reduction + double-shift QR for the
// values + inverse iteration for the nvec leading vectors (the dense route's pieces below): what
// block Arnoldi's Rayleigh-Ritz checks call since round 6 (0.3 ms at m = 64 where the complex
// Schur form takes 4.3 and the one-wavefront device kernel 4.0).
bool host_general_eig_fast(const double* a, int lda, int m, int nvec, double* wr, double* wi,
                           double* yre, double* yim, int ldy);

// ---- dense general eigenproblem of order n > 64: what follows the device's Hessenberg reduction
struct HostHessenberg {
  int n = 0;
  double norm = 0.0;            // max |h_ij|
  std::vector<double> H, V, tau;  // H: n x n upper Hessenberg; V: reflector k in row k (v[k+1] = 1)
};
// packed (n x n, row-major ld): Hessenberg matrix on and above the subdiagonal, reflector k in
// column k below it (LAPACK dgehd2's storage); false: a non-finite entry
bool host_hessenberg_unpack(const double* packed, size_t ld, int n, const double* tau,
                            HostHessenberg* w);
// all n eigenvalues (unordered); false: the QR iteration did not converge
bool host_hessenberg_eigenvalues(const HostHessenberg& w, double* wr, double* wi);
// eigenvectors of the original matrix for `count` eigenvalues, column-major (column q at q * ldv)
bool host_hessenberg_vectors(const HostHessenberg& w, const double* wr, const double* wi, int count,
                             double* vre, double* vim, size_t ldv, double* max_resid);

#endif  // SPECTRALCLUSTER_AMD_HOST_EIG_H_
