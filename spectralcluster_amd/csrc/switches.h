// Every environment switch the library reads, in one place.  None of them is a tuning knob:
//   trace switches   print the solvers' bookkeeping to stderr;
//   path switches    force a route the default configuration takes only for some inputs
//                    (repair / large-problem routes), so that tests can hold those routes to
//                    the same goldens: tests/test_gpu_alternate_paths.py runs every one.
// Each is read once per process.  (Round 2's experiment switches -- K-window throttle,
// staggered CU partners, bank priorities, a third bank, row caps -- are gone with the code
// they guarded; their measurements are in DESIGN_HISTORY.md and profiles/r02_*.)
#ifndef SPECTRALCLUSTER_AMD_SWITCHES_H_
#define SPECTRALCLUSTER_AMD_SWITCHES_H_

#include <cstdlib>
#include <cstring>

namespace sc {
namespace sw {

// SC_EIG_TRACE=1|2|3: block Lanczos / Arnoldi log (2: Ritz values, 3: scaling vectors)
inline int eig_trace() {
  static const int v = getenv("SC_EIG_TRACE") ? std::max(1, atoi(getenv("SC_EIG_TRACE"))) : 0;
  return v;
}
// SC_KMEANS_TRACE=1: seeds / iterations of the k-means stage
inline bool kmeans_trace() {
  static const bool v = getenv("SC_KMEANS_TRACE") != nullptr;
  return v;
}
// SC_GROUP_TRACE=1: per-group timeline of the grouped batch
inline bool group_trace() {
  static const bool v = getenv("SC_GROUP_TRACE") != nullptr;
  return v;
}
// SC_EIG_HOST_CHAIN=1: host-driven orthonormalisation chain (what the fused chain falls back
// to when a Krylov block is rank deficient)
inline bool eig_host_chain() {
  static const bool v = getenv("SC_EIG_HOST_CHAIN") != nullptr;
  return v;
}
// SC_EIG_DEVICE_RR=1: one-workgroup Jacobi for the Rayleigh-Ritz problems (the host solves them
// by default)
inline bool eig_device_rr() {
  static const bool v = getenv("SC_EIG_DEVICE_RR") != nullptr;
  return v;
}
// SC_EIG_FORCE_DENSE=1: straight to the dense landing pad (tridiagonalisation + bisection +
// inverse iteration) that otherwise takes over when block Lanczos gives up
inline bool eig_force_dense() {
  static const bool v = getenv("SC_EIG_FORCE_DENSE") != nullptr;
  return v;
}
// SC_MATVEC_SYM_MIN_N=<n>: upper-triangle block matvec from this size on (default 4096)
inline int matvec_sym_min_n() {
  static const int v = getenv("SC_MATVEC_SYM_MIN_N") ? atoi(getenv("SC_MATVEC_SYM_MIN_N")) : 4096;
  return v;
}
// SC_KMEANS_SINGLE=1: single-workgroup k-means (k > 32, other metrics, very large n)
inline bool kmeans_single() {
  static const bool v = getenv("SC_KMEANS_SINGLE") != nullptr;
  return v;
}
// SC_DIFFUSE=explicit|free|auto and SC_DIFFUSE_FREE_MIN_N=<n> (default 2048): route of a Diffuse
// that only feeds RowWiseNormalize / the Laplacian -- the fp64 product, or the matrix-free
// search of free_api.hip (sc_config.diffuse_mode / sc_set_diffuse_mode override).  Both routes
// are held to the same goldens (tests/test_gpu_diffuse_free.py, test_gpu_alternate_paths.py).
inline int diffuse_mode() {  // 0 auto, 1 explicit, 2 free
  static const int v = [] {
    const char* e = getenv("SC_DIFFUSE");
    if (!e) return 0;
    if (!strcmp(e, "explicit")) return 1;
    if (!strcmp(e, "free")) return 2;
    return 0;
  }();
  return v;
}
inline int diffuse_free_min_n() {
  static const int v =
      getenv("SC_DIFFUSE_FREE_MIN_N") ? atoi(getenv("SC_DIFFUSE_FREE_MIN_N")) : 2048;
  return v;
}
// (members of a grouped batch: the same switch, default 1536)
inline int diffuse_free_min_n_group() {
  static const int v =
      getenv("SC_DIFFUSE_FREE_MIN_N") ? atoi(getenv("SC_DIFFUSE_FREE_MIN_N")) : 1536;
  return v;
}
// SC_GEN_DENSE_MAX_N=<n> (default 512): eigengap requests on the general (non-symmetrisable) path
// up to this size take the dense Hessenberg route straight away (every eigenvalue to rounding
// level, 20-60 ms) instead of block Arnoldi (faster, but its decision-aware stop holds the values
// that cannot move the eigengap decision to 1e-3 only); 64 puts block Arnoldi back on everything
// above the one-wavefront solver
inline int gen_dense_max_n() {
  static const int v = getenv("SC_GEN_DENSE_MAX_N") ? atoi(getenv("SC_GEN_DENSE_MAX_N")) : 512;
  return v;
}
// SC_GROUP_QUANTIZE_PASS=1: the matrix-free members of a grouped front / sweep get their digits from
// a quantiser pass of their own (rounds 4-5) instead of from the grouped threshold pass
inline bool group_quantize_pass() {
  static const bool v = getenv("SC_GROUP_QUANTIZE_PASS") != nullptr;
  return v;
}
// SC_GEN_DEVICE_RR=1: the Rayleigh-Ritz problems of the narrow block Arnoldi (order <= 64) on the
// one-wavefront device kernel k_gen_eig (rounds 2-5) instead of the host
inline bool gen_device_rr() {
  static const bool v = getenv("SC_GEN_DEVICE_RR") != nullptr;
  return v;
}
// SC_GROUP_EQUAL_COUNT=1: the grouped batch cuts its size-sorted list into groups of equal
// COUNT dealt round-robin to the lanes (rounds 2-5) instead of groups of equal cost dealt
// longest-first (A/B measurements)
inline bool group_equal_count() {
  static const bool v = getenv("SC_GROUP_EQUAL_COUNT") != nullptr;
  return v;
}
// SC_NO_PREFETCH=1: the calls of a batch upload their embeddings themselves, one after the
// other (what a sequence of sc_predict calls does) instead of under their predecessor's pipeline
inline bool no_prefetch() {
  static const bool v = getenv("SC_NO_PREFETCH") != nullptr;
  return v;
}
// SC_FREE_NO_PRUNE=1: the digit product of the matrix-free Diffuse computes every tile (the
// skip list keeps them all): what an unstructured input gets anyway
inline bool free_no_prune() {
  static const bool v = getenv("SC_FREE_NO_PRUNE") != nullptr;
  return v;
}
// SC_GEN_LOOSE_BULK=1: rounds 3-5's stop rule of the general path (consumed eigenvalues that
// cannot move the eigengap decision held to 1e-3 instead of value_tol) -- A/B measurements only
inline bool gen_loose_bulk() {
  static const bool v = getenv("SC_GEN_LOOSE_BULK") != nullptr;
  return v;
}
// SC_SWEEP_ONE_BY_ONE=1: an AutoTune level as separate sc_eig_ncluster calls (what a level
// falls back to when member arenas do not fit or a value leaves the grouped path)
inline bool sweep_one_by_one() {
  static const bool v = getenv("SC_SWEEP_ONE_BY_ONE") != nullptr;
  return v;
}

}  // namespace sw
}  // namespace sc

#endif  // SPECTRALCLUSTER_AMD_SWITCHES_H_
