// Internal to the host side of the library (api.hip, eig_driver.hip, constraint_api.hip,
// callers_api.hip): the handle with its device arena, error plumbing, and the helpers those
// translation units share.  Not installed; the public surface is
// include/spectralcluster_amd.h.
#ifndef SPECTRALCLUSTER_AMD_HANDLE_H_
#define SPECTRALCLUSTER_AMD_HANDLE_H_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sc_internal.h"

using namespace sc;

// ------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

constexpr int kGroupBanks = 2;  // banks of member arenas of the grouped batch (groups in flight)
constexpr int kGroupLanes = 3;  // leads of the grouped batch (host threads, each with its banks)

// event slots of the stage timers of the current call (resolved once the stream has drained)
struct StageEvents {
  bool pending = false, fine = false, free_path = false;
  int begin = -1, after_refine = -1, after_scaling = -1, after_eig = -1;
  int diffuse[SC_MAX_OPS][2];
  int n_diffuse = 0;
  int blur[2] = {-1, -1}, thr[2] = {-1, -1};
};

struct sc_handle_s {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // current problem
  int n = 0, d = 0, ldn = 0, ldx = 0;
  bool have_x = false, have_affinity = false;
  bool have_cropval = false;  // cropval = CropDiagonal fill values of A0 (affinity GEMM epilogue)
  int n_vec = 0;          // eigenvector columns resident in E
  // matrices
  DevBuf X, Xn, A0, B1, B2;
  bool chain_flags_clean = false;    // flags[13..15] were cleared by the scaling kernel: the first
                                     // start of sym_topk skips its fill
  DevBuf Xalt;                       // second embeddings buffer: the NEXT call's upload lands here
  hipStream_t copy_stream = nullptr; // ... on this stream (predict_sequence, api.hip)
  // n-vectors
  DevBuf rowmax, rowsum, cvec, pvec, tvec, deg, dvec, cut, rmpart, splitk, tilemap;
  DevBuf cropval, statp;  // fused GEMM row statistics: result + per-tile partials
  // constraints: Cq (resident constraint matrix), Neumann-product work matrices, flag word
  DevBuf Cq, cp[5], symflag;
  bool have_constraint = false, constraint_symmetric = false, constraint_applied = false;
  bool affinity_symmetric = true;
  bool affinity_from_embeddings = false;  // symflag[1] then says whether a row was NaN
  int qn = 0;
  int tilemap_nt = 0;     // tile-grid size the resident tilemap was built for
  DevBuf tilemap_table;   // tile orders of every grid size <= kTilemapTableMax (ensure_tilemap)
  int tilemap_off[65] = {0};
  const int2* tilemap_cur = nullptr;  // what ensure_tilemap selected for the current n
  DevBuf blurw;           // device copy of the blur weights
  DevBuf blur_tmp;        // scratch of the two-pass blur of a radius above SC_MAX_BLUR_RADIUS
  std::vector<double> blur_ext;  // weights of such a radius (sc_set_blur_weights): 2 r + 1
  // eigen workspace
  DevBuf Q, Q2, Vs, W, partial, T, Y, Yt, theta, resid, G, Rinv, Hbuf, hsq, colnorm,
      flags;
  DevBuf E, Ek, Eio;      // eigenvectors (col-major), renormed copy, row-major I/O staging
  DevBuf mvsym;           // slabs of the symmetric block matvec
  // dense full-spectrum path (eig_dense.hip): d, e, all eigenvalues, reflector work vectors
  DevBuf td_d, td_e, td_theta, td_work, td_tau, td_panel;
  std::vector<double> spectrum;  // host copy: every eigenvalue of Op, descending
  std::vector<double> last_w;    // eigenvalues the last eig call consumed (reference order)
  // general (non-symmetric) eigen path: right scaling, Im(theta), complex Ritz vectors
  // (column-major), residual partials, restart codes, dense Laplacian scratch
  DevBuf crvec, thetai, Vre, Vim, gpart, gsrc, genL, gneg;
  const double* vs_scale = nullptr;  // Vs = vs_scale .* V in orthonormalize (default cvec)
  DevBuf ahc_size, ahc_chain, ahc_Z, ahc_lab, ahc_cent;  // size reduction (AHC) scratch
  DevBuf fb_part, fb_small, fb_x, fb_cent, fb_int;      // fallback decisions scratch
  // k-means workspace
  DevBuf kXc, kxsq, kclosest, kcand, kenorm, krnd, kcent, klab32, klab64, kinfo, kchain;
  DevBuf kbig, kbigw;     // more than kMaxVectors clusters: per-cluster arrays of k_kmeans<true>
  // pinned host scratch
  double* h_theta = nullptr;  // 3 * kLdq doubles (theta, resid, Im theta)
  int* h_flags = nullptr;
  double* h_rr = nullptr;     // pinned: T (kHostRR^2) | G (64) | Y (kHostRR^2): host Rayleigh-Ritz
  std::vector<sc_handle_s*> pool;  // extra handles (streams) of sc_predict_batch_streams
  // grouped batch (batch_group.hip): member arenas (own stream for the stages before the
  // eigensolver; the lockstep chains run on THIS handle's stream), the RandomState(0) doubles
  // of every k (k-means++ seeding)
  std::vector<sc_handle_s*> gslots;
  // last sc_eig_ncluster_sweep: member arena that holds value i's eigenvectors (-1: none),
  // what it reported, and the problem size it ran on (sc_sweep_adopt)
  std::vector<int> sweep_slot;
  std::vector<sc_diag> sweep_diags;
  std::vector<double> sweep_p;
  sc_config sweep_cfg;
  int sweep_n = 0;
  hipEvent_t sync_ev = nullptr;    // a member arena: "stages before the eigensolver done"
  DevBuf gkrnd;
  bool gkrnd_ready = false;
  // staging of the group's Rayleigh-Ritz checks (device + pinned host twins), the event the
  // host waits on, the k-means info words and labels of a group in one buffer each
  DevBuf gpack, gypack, ginfo, glabels;
  double* h_gpack = nullptr;
  double* h_gypack = nullptr;
  int* h_ginfo = nullptr;
  long long* h_glabels = nullptr;
  size_t h_glabels_count = 0;
  hipEvent_t gcheck_ev = nullptr;
  std::vector<sc_handle_s*> glanes;  // the leads of lanes 1.. (batch_group.hip)
  class HostPool* gpool = nullptr;  // host workers of the group checks (host_pool.h)
  // grouped front: the stages before the eigensolver of a whole group as grouped launches on
  // the stream of the group's bank, handed to this handle's stream through the bank's event
  hipStream_t gbank_stream[kGroupBanks] = {nullptr};
  hipStream_t gchain_stream = nullptr;  // the lockstep chains of a grouped batch (stands in for `stream`)
  hipEvent_t gbank_ev[kGroupBanks] = {nullptr};
  int gconv_hist[16] = {0};  // members of this batch that converged at basis 8 * index ...
  int gconv_seen = 0;        // ... of this many: where a speculative block is likely wasted
  hipEvent_t ev[48];
  int nev = 0;
  StageEvents stage_events;
  int profile_level = 1;  // sc_set_profiling: 0 totals only, 1 stages, 2 per-kernel events
  int mv_ev[16][2];       // event pairs around the block matvec launches (level 2)
  int n_mv_ev = 0;
  int aff_ev[2] = {-1, -1};  // around the affinity GEMM launch (level 2)
  // what the small per-call uploads last carried (skipped when unchanged)
  int blurw_radius = -1;
  double blurw_host[2 * SC_MAX_BLUR_RADIUS + 1];
  int krnd_k = -1, krnd_trials = -1;
  int kfirst_n = -1, kfirst = 0;  // first k-means++ centre of the last n (RandomState(0) draw)
  bool eig_skip_fused = false;  // next sym_topk: go straight to the host-driven chain
  // ---- matrix-free Diffuse (free_api.hip)
  int diffuse_mode = -1;   // sc_set_diffuse_mode: 0 auto, 1 explicit fp64 product, 2 matrix-free
                           // wherever the sequence allows it; -1: the environment's default
  DevBuf fq, ft32, fy1, fR, fscal, fwords, fcand, fY, fsplit, fypart, frpart;  // digits, T (fp32 tiles), A 1, sum|q|,
                           // scalars, M | count | ovf words, candidate lists, A Vs
  int free_prune = -1;     // sc_set_free_prune: 1 skip list on, 0 every tile, -1 environment
  DevBuf fq2part, fmx64, ftau64, fplan;  // tile skip list of the digit product (diffuse_free.hip):
                           // squared segment norms per (block, row), their maxima per 64-row
                           // group, the groups' thresholds, surviving tiles per tile row
  int* h_free = nullptr;   // pinned copy of the ovf words (80)
  bool free_on = false;    // the operator of the current solve is c .* A (A (c .* v)) + p .* v
  bool free_checked = false;  // ... and its overflow rows have been dealt with
  int free_lap = 0, free_rownorm = 0;  // what the scaling vectors were built for
  int free_ev[5] = {-1, -1, -1, -1, -1};  // event slots: begin | quantised | product | scans | stats
};

// hipEvent slots of the current call (reset by the entry points); -1 when exhausted
inline void ev_rec(sc_handle h, int* slot) {
  if (h->nev < 48) {
    hipEventRecord(h->ev[h->nev], h->stream);
    *slot = h->nev++;
  } else {
    *slot = -1;
  }
}
inline float ev_ms(sc_handle h, int a, int b) {
  if (a < 0 || b < 0) return 0.f;
  float ms = 0.f;
  hipEventElapsedTime(&ms, h->ev[a], h->ev[b]);
  return ms;
}

constexpr int kTilemapTableMax = 64;
constexpr int kMaxCols = 128;  // eigenvector columns the arena can hold
// Leading dimension of the n x n matrices.  A row stride that is a multiple of 4 KiB maps
// the 128 rows of an operand panel onto the same few L2 sets (the GEMM reads one 128-byte
// line per row and K-tile): such strides get one extra 128-byte line.
inline int matrix_ld(int n) {
  int ld = round_up(n, 16);
  if (ld % 512 == 0) ld += 16;
  return ld;
}

// eigenvectors are column-major on the device: column j at E + j * lde, lde = round_up(n, 16)

#define SC_HIP(h, call)                                                         \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
      return e_ == hipErrorOutOfMemory ? SC_ERR_OOM : SC_ERR_HIP;               \
    }                                                                           \
  } while (0)

#define SC_TRY(expr)                \
  do {                              \
    int rc_ = (expr);               \
    if (rc_ != SC_OK) return rc_;   \
  } while (0)

inline int fail(sc_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

inline int grow(sc_handle h, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes) return SC_OK;
  if (b.p) {
    SC_HIP(h, hipStreamSynchronize(h->stream));
    SC_HIP(h, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  SC_HIP(h, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return SC_OK;
}
template <typename T>
inline T* ptr(const DevBuf& b) {
  return reinterpret_cast<T*>(b.p);
}

// `affinity_copy` false: no A0 (a member arena of an AutoTune sweep only holds B1 / B2)
int ensure_matrices(sc_handle h, int n, int d, bool affinity_copy = true);

// (ti, tj) order of the symmetric GEMM tiles for problems of n rows (cached per handle)
int ensure_tilemap(sc_handle h, int n);
// `count` independent calls (indices idx[0..count) into xs / ns / labels / diags; idx == nullptr:
// 0..count-1) one after the other on handle h, each call's upload under its predecessor's pipeline
int predict_sequence(sc_handle h, const int* idx, int count, const double* const* xs,
                     const int* ns, int d, const sc_config* cfg, int64_t* const* labels,
                     sc_diag* diags);

int ensure_eig(sc_handle h, int n);

int ensure_gen(sc_handle h, int n);

// k: clusters the k-means stage will be asked for (buffers grow past the 64-cluster default)
int ensure_kmeans(sc_handle h, int n, int k = kMaxVectors);
int ensure_vectors(sc_handle h, int n, int cols);

inline int check_last(sc_handle h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    h->err = std::string(what) + ": " + hipGetErrorString(e);
    return SC_ERR_HIP;
  }
  return SC_OK;
}


// ---- shared between the translation units -------------------------------------------
int h2d_matrix(sc_handle h, const double* src, int rows, int cols, double* dst, int ld);
int d2h_matrix(sc_handle h, const double* src, int ld, int rows, int cols, double* dst);
int validate_config(sc_handle h, const sc_config* cfg);
// utils.compute_number_of_clusters on a host array (api.hip)
void eigengap_core(const double* w, int count, int max_clusters, double stop_eigenvalue,
                   int eigengap_type, int descend, double wmax, int* n_clusters,
                   double* max_delta);

// eigen drivers (eig_driver.hip)
struct EigRequest {
  int descend;          // 1: report largest first (w = theta); 0: w = -theta ascending
  int max_clusters;     // 0 = None
  int min_clusters;     // 0 = None
  double stop_eigenvalue;
  int eigengap_type;
  int use_stop;         // stop_eigenvalue only on the descending branch
  double value_tol, vector_tol;
  int max_cycles;
  int fixed_count;      // > 0: plain "count extreme eigenpairs" request (stage API)
  // General path only.  Consumed eigenvalues deep in a dense bulk converge arbitrarily
  // slowly in a small Krylov basis, yet cannot influence the result: only the two values
  // that form the maximum gap (and the normaliser of NormalizedDiff) are held to value_tol;
  // the others must be accurate enough that, with their residual intervals, no other gap
  // can reach the maximum and no comparison with stop_eigenvalue can flip.
  int decision_aware = 0;
  // General path, internal (gen_topk's far-end solve, eig_driver.hip): the operator with the
  // opposite sign -- the eigenvalues of largest real part of +L instead of -L; give up with
  // SC_ERR_NOT_CONVERGED instead of landing on the dense route; and, for the main solve that
  // follows, np.max(eigenvalues) of the ascending NormalizedDiff rule as that solve found it.
  int negate = 0;
  int no_dense = 0;
  int have_far = 0;
  double far_value = 0.0;
};

struct EigDecision {
  bool enough = false;     // basis large enough to take a decision
  bool converged = false;
  int kw = 0;              // eigenvalues reported
  int kvec = 0;            // vectors that must be accurate
  int n_clusters_raw = 0;
  double max_delta = 0.0;
  double max_resid = 0.0;
  bool unsupported = false;
  int fail_kind = 0;       // 1 consumed value, 2 far-end value, 3 vector, 4 decision (trace)
  int fail_index = -1;
};

// where eig_ncluster_impl(front_only) left the refined matrix
struct FrontResult {
  const double* matrix = nullptr;
  double* scratch = nullptr;
  int ld = 0;
  bool symmetric = false, folded_rownorm = false;
  // matrix-free Diffuse: `matrix` is the symmetric A BEFORE Diffuse (the operator applies it
  // twice), the row statistics of S = A A^T are in the handle, its overflow record in h_free
  bool free_op = false;
};
EigRequest make_eig_request(const sc_config* cfg);
int upload_blur_weights(sc_handle h, const sc_config* cfg);  // into h->blurw, on h->stream
// `resume`: the stages before the eigensolver already ran (a FrontResult of this handle)
int eig_ncluster_impl(sc_handle h, const sc_config* cfg, sc_diag* diag, FrontResult* front_only,
                      const FrontResult* resume = nullptr, bool defer_timing = false);
// RandomState(0) stream of the k-means seeding: the uniform that picks the first centre, the
// trial count 2 + int(log k), and the (k - 1) * trials doubles after it (api.hip)
void kmeans_seed_constants(int k, double* u_first, int* trials, std::vector<double>* rnd);
KmeansWorkspace kmeans_workspace(sc_handle h);

// One member of a lockstep group solve (eig_driver.hip: sym_topk_group).
struct GroupEigMember {
  sc_handle h = nullptr;      // arena of the member (its matrix ready on lead->stream)
  const double* S = nullptr;  // refined symmetric matrix, scaling vectors resident in h
  int ld = 0, n = 0;
  EigRequest rq;
  // results
  int status = 0;             // 0 solved; 1 needs the single-call path (rare branch taken)
  EigDecision dc;
  std::vector<double> w;      // consumed eigenvalues (reference order)
  int basis = 0, passes = 0;
  // matrix-free Diffuse: S is the symmetric A before Diffuse, the operator is
  // diag(p) + diag(c) A A diag(c) (two block products per step, through h->fY)
  bool free_op = false;
};
// Block Lanczos of up to kGroupMax members in lockstep: one launch per chain link / matvec
// for the whole group, one host synchronisation per Rayleigh-Ritz check for the whole group.
// Leaves Ritz vectors in every solved member's h->E (h->n_vec set).
// `want_vectors` false: eigenvalues and the eigengap decision only (AutoTune sweep).
int sym_topk_group(sc_handle lead, GroupEigMember* mem, int count, bool want_vectors = true);
// `any_size`: also n >= 4096 (the group takes the upper-triangle matvec there)
bool sym_group_eligible(int n, const EigRequest& rq, bool any_size = false);

// `scratch`: a free n x ld matrix (the dense full-spectrum path materialises Op there)
int sym_topk(sc_handle h, const double* S, int ld, int n, const EigRequest& rq, sc_diag* diag,
             EigDecision* out_dc, std::vector<double>* out_w, double* scratch);
// `scratch`: a free n x ld matrix (the dense Hessenberg route reduces a copy of M there; may be
// null for n <= 64)
int gen_topk(sc_handle h, const double* M, int ld, int n, int laplacian_type,
             const EigRequest& rq, sc_diag* diag, EigDecision* out_dc,
             std::vector<double>* out_w, double* scratch);

// matrix-free Diffuse (free_api.hip)
// does this call take the matrix-free route for a Diffuse whose output only feeds
// RowWiseNormalize / the Laplacian?  (mode of the handle, problem size, request)
bool free_diffuse_wanted(sc_handle h, const sc_config* cfg, int n, const EigRequest& rq,
                         bool in_group = false);
// enqueue the statistics of S = A A^T (h->rowmax, h->rowsum) on h->stream; no synchronisation.
// The overflow words travel to h->h_free behind them.
// `have_amax`: h->fscal[0] already holds max|a| (or an upper bound of it) for this A
int free_group_begin(sc_handle* hs, const double* const* A, const double* const* cuts,
                     const double* ps, int count, const int* lds, const int* ns, hipStream_t s,
                     double floor_value, struct FreeItem* items);
int free_group_end(sc_handle* hs, const struct FreeItem* items, int count, hipStream_t s);
// (the grouped threshold pass writes the digits: free_api.hip)
int free_group_prepare(sc_handle* hs, const double* const* A, const double* const* cuts,
                       const double* ps, int count, const int* lds, const int* ns, hipStream_t s,
                       double floor_value, struct FreeItem* items, struct TsDigits* digits);
int free_group_digits(sc_handle* hs, const struct FreeItem* items, int count, hipStream_t s);
int free_fused_prepare(sc_handle h, hipStream_t s, int n, const double* cut, double p,
                       double floor_value);
int free_diffuse_stats(sc_handle h, const double* A, int ld, int n, bool have_amax = false,
                       bool digits_ready = false);
int ensure_free(sc_handle h, int n);
// the same pipeline in pieces on a given stream (the AutoTune sweep runs the digit product of
// all its members as one grouped launch between them): max|a| + digits | ... | candidates +
// exact statistics + the overflow words on their way to h->h_free
int free_product(sc_handle h, hipStream_t s, int n);
int free_stats_begin(sc_handle h, hipStream_t s, const double* A, int ld, int n, bool have_amax);
int free_stats_end(sc_handle h, hipStream_t s, const double* A, int ld, int n, bool timed,
                   const int* plan = nullptr);
// after the stream has drained: rows with more candidates than the cap are evaluated in full.
// *changed: rowmax was rewritten (the scaling vectors must be rebuilt); *too_many: more such
// rows than the exact route takes (the caller forms S explicitly).
int free_fix_overflow(sc_handle h, const double* A, int ld, int n, bool* changed, bool* too_many);
// W = p .* V + c .* (A (A Vs)) through h->fY (both halves on h->stream)
void free_apply_operator(sc_handle h, const double* A, int ld, int n, bool sym_mv,
                         const double* V, int ldv);
bool wants_full_spectrum(const EigRequest& rq);

// constraints (constraint_api.hip)
int device_is_symmetric(sc_handle h, const double* m, int n, int ld, bool* out);
int adjust_affinity(sc_handle h, const sc_config* cfg, const double* a, bool sym_a, double* out,
                    int n, int ld);
bool constraint_active(sc_handle h, const sc_config* cfg, bool before);

#endif  // SPECTRALCLUSTER_AMD_HANDLE_H_
