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

#endif  // SPECTRALCLUSTER_AMD_HOST_EIG_H_
