/*
 * spectralcluster_amd.h -- C ABI of the MI355X (gfx950) implementation of the
 * dense hot path of SpectralClusterer.predict() (wq2012/SpectralCluster
 * v0.2.22).  Plain pointers and sizes only; no torch / numpy types.
 *
 * Every entry point names the reference interface it replaces (paths are
 * relative to the reference checkout, `spectralcluster/...`).  The reference
 * is pure Python, so "the binding a maintainer would add" is a ctypes stub;
 * see INTEGRATION.md.
 *
 * Conventions
 *   - all matrices are float64, C-contiguous row-major, caller-owned;
 *   - every function returns an sc_status (0 = ok, < 0 = error); the text
 *     of the last error of a handle is available from sc_last_error();
 *   - one handle <-> one device <-> one HIP stream.  A handle is NOT
 *     thread-safe; distinct handles are independent;
 *   - no host pointer is retained after a call returns.
 */
#ifndef SPECTRALCLUSTER_AMD_H_
#define SPECTRALCLUSTER_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 7
#define SC_MAX_OPS 16
#define SC_MAX_BLUR_RADIUS 32
#define SC_MAX_EIG 128 /* max eigenvalues reported in sc_diag */
#define SC_MAX_STAGES 16

typedef struct sc_handle_s* sc_handle;

typedef enum sc_status {
  SC_OK = 0,
  SC_ERR_INVALID = -1,       /* bad argument (maps to ValueError/TypeError) */
  SC_ERR_OOM = -2,           /* device allocation failed */
  SC_ERR_HIP = -3,           /* HIP runtime error */
  SC_ERR_NOT_CONVERGED = -4, /* eigensolver did not reach tolerance */
  SC_ERR_UNSUPPORTED = -5,   /* configuration outside the device path */
  SC_ERR_NON_FINITE = -6     /* NaN / inf reached the eigen stage (numpy: LinAlgError) */
} sc_status;

/* refinement.py:11-18 RefinementName */
typedef enum sc_op {
  SC_OP_CROP_DIAGONAL = 1,
  SC_OP_GAUSSIAN_BLUR = 2,
  SC_OP_ROW_WISE_THRESHOLD = 3,
  SC_OP_SYMMETRIZE = 4,
  SC_OP_DIFFUSE = 5,
  SC_OP_ROW_WISE_NORMALIZE = 6
} sc_op;

/* refinement.py:21-27 ThresholdType, :30-36 SymmetrizeType */
enum { SC_THRESHOLD_ROW_MAX = 1, SC_THRESHOLD_PERCENTILE = 2 };
enum { SC_SYMMETRIZE_MAX = 1, SC_SYMMETRIZE_AVERAGE = 2 };
/* laplacian.py:9-21 LaplacianType (0 = laplacian_type None) */
enum {
  SC_LAPLACIAN_NONE = 0,
  SC_LAPLACIAN_AFFINITY = 1,
  SC_LAPLACIAN_UNNORMALIZED = 2,
  SC_LAPLACIAN_RANDOM_WALK = 3,
  SC_LAPLACIAN_GRAPH_CUT = 4
};
/* constraint.py:10-16 ConstraintName (0 = constraint_options None), :19-22 IntegrationType */
enum {
  SC_CONSTRAINT_NONE = 0,
  SC_CONSTRAINT_AFFINITY_INTEGRATION = 1,
  SC_CONSTRAINT_PROPAGATION = 2
};
enum { SC_INTEGRATION_MAX = 1, SC_INTEGRATION_AVERAGE = 2 };
/* linkage of the size-reduction / fallback agglomerative clustering */
enum { SC_LINKAGE_COMPLETE = 1, SC_LINKAGE_AVERAGE = 2 };
/* custom_dist of run_kmeans (custom_distance_kmeans.py:13-52): the scipy cdist metric
 * names that run on the device.  (A falsy custom_dist makes the reference call .predict()
 * on an unfitted sklearn KMeans, :33-36/:51 -- it always raises; the host side mirrors that.) */
enum {
  SC_KMEANS_COSINE = 0,
  SC_KMEANS_EUCLIDEAN = 1,
  SC_KMEANS_SQEUCLIDEAN = 2,
  SC_KMEANS_CITYBLOCK = 3,
  SC_KMEANS_CHEBYSHEV = 4,
  SC_KMEANS_CORRELATION = 5, /* cosine distance of the row-centred operands */
  SC_KMEANS_BRAYCURTIS = 6,
  SC_KMEANS_CANBERRA = 7
};
/* utils.py:10-17 EigenGapType */
enum { SC_EIGENGAP_RATIO = 1, SC_EIGENGAP_NORMALIZED_DIFF = 2 };

/* Which eigen path ran (sc_diag.eig_path) */
enum {
  SC_EIG_PATH_DENSE_JACOBI = 1,   /* symmetric, n <= 128 */
  SC_EIG_PATH_BLOCK_LANCZOS = 2,  /* symmetric, larger n */
  SC_EIG_PATH_DENSE_GENERAL = 3,  /* non-symmetric, n <= 64: Hessenberg + complex QR */
  SC_EIG_PATH_BLOCK_ARNOLDI = 4,  /* non-symmetric, larger n */
  /* symmetric, n > 128, every eigenvalue consumed (max_clusters=None with a Laplacian,
   * or the ascending NormalizedDiff gap's np.max): Householder tridiagonalisation +
   * Sturm bisection for the values, block Lanczos for the few vectors k-means takes */
  SC_EIG_PATH_DENSE_TRIDIAG = 5,
  /* block Lanczos gave up (sc_diag.eig_fallback says why): values as above and the vectors by
   * inverse iteration on the tridiagonal form + the Householder back-transform.  Always
   * returns, like np.linalg.eig (utils.py:59) */
  SC_EIG_PATH_DENSE_FULL = 6,
  /* non-symmetric, n > 64, when more eigenvalues are read than a block Arnoldi basis holds
   * (max_clusters=None with a Laplacian, max_clusters > 63, min_clusters > 64) or block Arnoldi
   * gives up: Householder reduction to Hessenberg form on the device, every eigenvalue by the
   * double-shift QR iteration and the eigenvectors k-means reads by inverse iteration on the
   * host.  Always returns, like np.linalg.eig (utils.py:59) */
  SC_EIG_PATH_DENSE_HESSENBERG = 7
};

/* Stage slots of sc_diag.stage_ms */
enum {
  SC_STAGE_AFFINITY = 0,
  SC_STAGE_REFINE = 1,    /* all refinement ops except Diffuse */
  SC_STAGE_DIFFUSE = 2,
  SC_STAGE_SCALING = 3,   /* row stats + Laplacian/normalise scaling vectors */
  SC_STAGE_EIG = 4,
  SC_STAGE_KMEANS = 5,
  SC_STAGE_TOTAL = 6,
  /* per-kernel slots, filled only at sc_set_profiling(h, 2) */
  SC_STAGE_BLUR = 7,           /* CropDiagonal + GaussianBlur kernel */
  SC_STAGE_THRESHOLD_SYM = 8,  /* RowWiseThreshold + Symmetrize kernel */
  SC_STAGE_MATVEC = 9,         /* sum over the block matvec launches of the eigen stage */
  SC_STAGE_AFFINITY_GEMM = 10, /* the affinity GEMM launch alone */
  /* matrix-free Diffuse (sc_diag.diffuse_path == SC_DIFFUSE_PATH_FREE); their sum is
   * SC_STAGE_DIFFUSE of such a call */
  SC_STAGE_FREE_QUANTIZE = 11, /* max|a| + 8-bit digits of A, rowsum(A) */
  SC_STAGE_FREE_PRODUCT = 12,  /* exact integer MFMA product of the digits */
  SC_STAGE_FREE_SCAN = 13,     /* row maxima + candidates within the proven slack */
  SC_STAGE_FREE_STATS = 14     /* exact fp64 rowmax(S) of the candidates, rowsum(S) */
};

/* How the last Diffuse (refinement.py:229-234) of the sequence ran (sc_diag.diffuse_path) */
enum {
  SC_DIFFUSE_PATH_NONE = 0,
  SC_DIFFUSE_PATH_EXPLICIT = 1,           /* S = A A^T by the fp64 MFMA GEMM */
  /* S never formed: rowmax(S) by the digit product + exact recheck, S V = A (A V) */
  SC_DIFFUSE_PATH_FREE = 2,
  /* started matrix-free, S formed after all (a dense eigen route read entries, or more rows
   * than the exact-row route takes had to be evaluated in full) */
  SC_DIFFUSE_PATH_FREE_THEN_EXPLICIT = 3
};

/*
 * POD mirror of SpectralClusterer.__init__ (spectral_clusterer.py:29-46) and
 * RefinementOptions (refinement.py:76-100), restricted to the hot path.
 * 0 means "None" for min_clusters / max_clusters.
 */
typedef struct sc_config {
  int32_t n_ops;                 /* len(refinement_sequence), 0 = no refinement */
  int32_t ops[SC_MAX_OPS];       /* sc_op values, applied in order */
  /* GaussianBlur: the 2*radius+1 symmetric weights as scipy computes them
   * (sigma -> radius int(4 sigma + .5)); radius 0 = plain copy (sigma == 0).
   * sc_gaussian_weights() fills these from sigma. */
  int32_t blur_radius;
  double blur_weights[2 * SC_MAX_BLUR_RADIUS + 1];
  double p_percentile;           /* refinement.py:79 */
  double soft_multiplier;        /* refinement.py:83 */
  int32_t threshold_type;        /* SC_THRESHOLD_* */
  int32_t binarize;              /* thresholding_with_binarization */
  int32_t preserve_diagonal;     /* thresholding_preserve_diagonal */
  int32_t symmetrize_type;       /* SC_SYMMETRIZE_* */
  int32_t laplacian_type;        /* SC_LAPLACIAN_* */
  int32_t min_clusters;          /* 0 = None */
  int32_t max_clusters;          /* 0 = None */
  double stop_eigenvalue;        /* spectral_clusterer.py:36 */
  int32_t eigengap_type;         /* SC_EIGENGAP_* */
  int32_t row_wise_renorm;       /* spectral_clusterer.py:37 */
  int32_t max_iter;              /* custom k-means iterations (:39) */
  /* Eigensolver knobs (no reference equivalent). 0 selects the default. */
  double eig_value_tol;          /* residual bound / |eigenvalue| on consumed values (1e-6) */
  double eig_vector_tol;         /* residual tol, relative to ||M||, on the
                                    eigenvectors handed to k-means (1e-10) */
  int32_t eig_max_cycles;        /* restart cycles block Lanczos may spend before the dense
                                    eigensolver takes over (0: default 40; < 0: none) */
  /* ConstraintOptions (constraint.py:26-48); used only while a constraint matrix
   * is resident (sc_set_constraint), like constraint_matrix=None in the reference */
  int32_t constraint_name;       /* SC_CONSTRAINT_* */
  int32_t constraint_before_refinement; /* apply_before_refinement */
  int32_t integration_type;      /* SC_INTEGRATION_* */
  double constraint_alpha;       /* constraint_propagation_alpha (0.6) */
  int32_t kmeans_metric;         /* SC_KMEANS_* (custom_dist, spectral_clusterer.py:38) */
  /* Route of a Diffuse that only feeds RowWiseNormalize / the Laplacian (no reference
   * equivalent; see sc_set_diffuse_mode): 0 = the handle's / process default, 1 = explicit fp64
   * product, 2 = matrix-free wherever the sequence allows it */
  int32_t diffuse_mode;
  int32_t reserved[4];
} sc_config;

typedef struct sc_diag {
  int32_t n;                     /* problem size */
  int32_t n_clusters_raw;        /* eigengap result before max(., min_clusters) */
  int32_t n_clusters;            /* value handed to k-means */
  int32_t eig_path;              /* SC_EIG_PATH_* */
  double max_delta;              /* max eigengap (utils.py:74-130, 2nd result) */
  int32_t n_eigenvalues;         /* entries valid in eigenvalues[] */
  int32_t eig_descending;        /* 1: largest first (None/Affinity); 0: smallest first */
  double eigenvalues[SC_MAX_EIG];/* in the order compute_sorted_eigenvectors returns */
  int32_t eig_matvec_passes;     /* passes over the n x n operator */
  int32_t eig_block;             /* vectors per pass */
  int32_t eig_basis;             /* Krylov basis size at exit */
  int32_t eig_cycles;            /* restart cycles used */
  double eig_max_residual;       /* max residual norm over accepted Ritz pairs */
  int32_t kmeans_iterations;     /* cosine k-means distance passes */
  int32_t symmetry_state;        /* 1 SYM, 2 DIAG*SYM (after RowWiseNormalize), 3 GENERAL */
  int32_t eig_host_chain;        /* 1: the fused Lanczos chain met a rank-deficient block and
                                    the host-driven repair chain redid the solve */
  int32_t eig_fallback;          /* 0, or why block Lanczos handed over to the dense path:
                                    1 restart budget spent, 2 projected eigenproblem failed,
                                    3 no full-rank Krylov block, 4 forced (SC_EIG_FORCE_DENSE),
                                    5 more than 64 eigenvectors wanted (> 64 selected clusters),
                                    7 eight equal Ritz values ahead of the decisive gap (an
                                    eigenvalue of multiplicity > 8: the dense path counts it);
                                    on the general path (eig_path 7) also 6: every eigenvalue
                                    is read (max_clusters=None with a Laplacian), 8: n <= 512,
                                    where the dense route is the default, 9: ascending
                                    NormalizedDiff reads np.max(eigenvalues), the far end of
                                    the spectrum (utils.py:110,123) */
  float stage_ms[SC_MAX_STAGES]; /* hipEvent time per SC_STAGE_* slot */
  int32_t diffuse_path;          /* SC_DIFFUSE_PATH_* */
  int32_t free_candidates;       /* matrix-free Diffuse: exact dot products evaluated (n + few) */
  int32_t free_overflow_rows;    /* ... rows evaluated in full (more candidates than the cap) */
  int32_t free_tiles_run;        /* ... 128 x 128 tiles of the digit product that were computed:
                                    the others -- of ceil(n/128) (ceil(n/128) + 1) / 2 -- were
                                    excluded by the segment-norm bound */
} sc_diag;

/* ---- library / device ---------------------------------------------------- */
int sc_abi_version(void);
/* sizeof(sc_config) / sizeof(sc_diag) as compiled, so a binding can verify its mirror */
int sc_struct_sizes(int* config_bytes, int* diag_bytes);
/* number of visible HIP devices (0 if none / runtime unavailable) */
int sc_device_count(void);
/* copies the device name (e.g. "AMD Instinct MI355X") and gcnArchName */
int sc_device_info(int device, char* name, int name_len, char* arch, int arch_len,
                   int* compute_units, int64_t* total_mem_bytes);

/* ---- handle -------------------------------------------------------------- */
int sc_create(int device, sc_handle* out);
int sc_destroy(sc_handle h);
/* pre-size the device arena for problems up to (n_max, d_max); optional */
int sc_reserve(sc_handle h, int n_max, int d_max);
const char* sc_last_error(sc_handle h);
int sc_synchronize(sc_handle h);
/* hipEvent timers in sc_diag.stage_ms: 1 (default) = one pair per stage, 2 = additionally
 * around the individual hot kernels (bench.py's per-kernel roofline list) */
int sc_set_profiling(sc_handle h, int level);
/* Route of a Diffuse (refinement.py:229-234) that is followed only by RowWiseNormalize and the
 * Laplacian -- the ICASSP2018 sequence.  0 (default): matrix-free from n = 2048 on (S = A A^T is
 * never formed: rowmax(S) from an exact 8-bit-digit integer MFMA product + fp64 recheck of the
 * candidates within a proven slack, the eigensolver applies A twice); 1: always the explicit
 * fp64 MFMA product; 2: matrix-free wherever the sequence allows it (n > 128); -1: back to
 * the process default (environment SC_DIFFUSE=explicit|free|auto, SC_DIFFUSE_FREE_MIN_N).
 * Both routes return the same rowmax / rowsum to summation order; sc_diag.diffuse_path says
 * which one ran. */
int sc_set_diffuse_mode(sc_handle h, int mode);
/* Tile skip list of the matrix-free route's digit product (ABI 7): 1 (default) = tiles that a
 * Cauchy-Schwarz bound on 64-column digit-segment norms proves free of row maxima and candidates
 * are not computed (exact for any input: rowmax / rowsum / candidate sets are those of the full
 * product); 0 = every tile (A/B measurements, parity tests); -1 = the process default
 * (environment SC_FREE_NO_PRUNE).  sc_diag.free_tiles_run reports what ran. */
int sc_set_free_prune(sc_handle h, int on);

/* fills cfg with the reference defaults (refinement.py:76-100,
 * spectral_clusterer.py:29-46): no ops, sigma 1 weights, p .95, mult .01 ... */
int sc_config_default(sc_config* cfg);
/* scipy.ndimage _gaussian_kernel1d(sigma, order 0, radius int(4 sigma + .5)) */
int sc_gaussian_weights(double sigma, int32_t* radius, double* weights);
/* GaussianBlur with sigma > 8 (radius > SC_MAX_BLUR_RADIUS; refinement.py:154-162 has no
 * limit): its 2 * radius + 1 weights do not fit sc_config -- upload them once, they stay
 * resident in the handle, and a config with blur_radius == radius uses them.  Such a config
 * carries the CENTRAL 2 * SC_MAX_BLUR_RADIUS + 1 weights (weights[radius - 32 .. radius + 32])
 * in cfg->blur_weights: every call checks them against the resident ones, so two sigmas that
 * share a radius cannot be confused (SC_ERR_UNSUPPORTED: upload again). */
int sc_set_blur_weights(sc_handle h, int radius, const double* weights);

/* ---- whole path ---------------------------------------------------------- */
/*
 * SpectralClusterer.predict (spectral_clusterer.py:201-314) for the in-scope
 * branch: affinity -> refinement -> Laplacian -> top-k eigen + eigengap ->
 * cosine k-means.  X: (n, d).  labels: n int64 (caller-allocated).
 */
int sc_predict(sc_handle h, const double* x, int n, int d, const sc_config* cfg,
               int64_t* labels, sc_diag* diag);

/* Split form, used for device-resident timing and by AutoTune:            */
/* H2D of the embeddings into the handle's arena. */
int sc_set_embeddings(sc_handle h, const double* x, int n, int d);
/* utils.compute_affinity_matrix (utils.py:20-41) on the resident embeddings. */
int sc_compute_affinity(sc_handle h);
/* H2D of a caller-supplied (n, n) affinity (custom affinity_function /
 * _compute_eigenvectors_ncluster(affinity), spectral_clusterer.py:108). */
int sc_set_affinity(sc_handle h, const double* affinity, int n);
/*
 * SpectralClusterer._compute_eigenvectors_ncluster (spectral_clusterer.py:
 * 108-168) on the resident affinity, which is left untouched (AutoTune calls
 * this once per p_percentile, spectral_clusterer.py:274-287).  Eigenvectors
 * stay on the device; diag receives eigenvalues, n_clusters_raw, max_delta.
 */
int sc_eig_ncluster(sc_handle h, const sc_config* cfg, sc_diag* diag);
/* The eigenvalues the last sc_eig_ncluster / sc_predict consumed, in the order
 * compute_sorted_eigenvectors returns them (utils.py:62-70): sc_diag.eigenvalues holds at
 * most SC_MAX_EIG of them, these calls give all (n of them with max_clusters=None). */
int sc_num_eigenvalues(sc_handle h);
int sc_get_eigenvalues(sc_handle h, double* out, int count);
/* number of eigenvector columns currently resident */
int sc_num_eigenvectors(sc_handle h);
/* D2H of the first ncols resident eigenvectors as an (n, ncols) matrix */
int sc_get_eigenvectors(sc_handle h, double* out, int n, int ncols);
/*
 * predict() tail (spectral_clusterer.py:295-313): slice [:, :n_clusters],
 * optional row re-norm, run_kmeans (custom_distance_kmeans.py:13-52).
 */
int sc_cluster(sc_handle h, const sc_config* cfg, int n_clusters,
               int64_t* labels, sc_diag* diag);
/*
 * Constraints (constraint.py:95-164; spectral_clusterer.py:137-142, 259-264).
 * sc_set_constraint uploads the (n, n) constraint matrix (it stays resident until
 * sc_clear_constraint; exact symmetry is detected on the device).  While one is
 * resident and cfg->constraint_name != 0:
 *   - apply_before_refinement: sc_predict / sc_run_resident adjust the affinity right
 *     after computing it; on the split path call sc_apply_constraint once after
 *     sc_compute_affinity / sc_set_affinity (it rewrites the resident affinity);
 *   - otherwise sc_eig_ncluster adjusts the refined matrix (each call).
 * ConstraintPropagation's (I - alpha A_norm)^-1 is evaluated as the Neumann product
 * prod_j (I + (alpha A_norm)^(2^j)) with fp64 MFMA GEMMs, which needs |alpha| < 1 and
 * a non-negative affinity (spectral radius of A_norm <= 1) -- both hold for the
 * reference's cosine affinity and its 0.4 / 0.6 presets.
 */
int sc_set_constraint(sc_handle h, const double* constraint_matrix, int n);
int sc_clear_constraint(sc_handle h);
int sc_apply_constraint(sc_handle h, const sc_config* cfg);
/* affinity + eig_ncluster + cluster on the resident embeddings (no H2D of X) */
int sc_run_resident(sc_handle h, const sc_config* cfg, int64_t* labels,
                    sc_diag* diag);
/* one AutoTune search level (autotune.py:98-111): sc_eig_ncluster for `count` values of
 * p_percentile on the resident affinity; diags[i] reports what sc_eig_ncluster would for
 * p_values[i].  The values of a level differ only in the row threshold: the stages after it
 * run as grouped launches, the eigensolvers in lockstep.  Leaves no eigenvectors resident
 * in the handle itself (sc_sweep_adopt fetches a value's). */
int sc_eig_ncluster_sweep(sc_handle h, const sc_config* cfg, const double* p_values, int count,
                          sc_diag* diags);
/* The eigenvectors of value `index` of the last sweep become the resident ones -- what
 * sc_eig_ncluster with p_values[index] would leave, without evaluating the winner a second
 * time (the reference's search keeps the winner's eigenvectors, spectral_clusterer.py:
 * 274-292).  cfg: the sweep's configuration with p_percentile = p_values[index] (checked).
 * SC_ERR_UNSUPPORTED when that value's solve left the grouped path or the configuration is
 * not the sweep's (then call sc_eig_ncluster).  Follow with sc_cluster. */
int sc_sweep_adopt(sc_handle h, const sc_config* cfg, int index, sc_diag* diag);
/* a Python `for` over predict() in the reference: count
 * independent utterances, xs[i] is (ns[i], d); labels[i] has ns[i] slots. */
int sc_predict_batch(sc_handle h, const double* const* xs, const int* ns, int d,
                     int count, const sc_config* cfg, int64_t* const* labels,
                     sc_diag* diags);
/* the same over `streams` HIP streams of the handle's device (pooled arenas, one host
 * thread per stream inside the call, longest utterances first): small utterances cannot
 * fill the GPU and their pipeline is latency-bound, several in flight overlap */
int sc_predict_batch_streams(sc_handle h, const double* const* xs, const int* ns, int d,
                             int count, const sc_config* cfg, int64_t* const* labels,
                             sc_diag* diags, int streams);
/* the same as groups of `group` (<= 16) utterances per launch: the stages before the
 * eigensolver are enqueued member after member (one launch for all members' GEMM tiles when
 * the sequence is the ICASSP2018 one), the block Lanczos chain and the k-means chain of the
 * members advance in lockstep (one launch per step and one host synchronisation per check for
 * the whole group).  The groups are dealt to up to three lanes -- internal lead handles with
 * their own streams, member arenas and host thread -- so up to three groups' chains are in
 * flight; the call returns when all lanes are done.  Per-utterance results agree with
 * sc_predict to the solver's tolerance (whole-K tile sums where a short single call splits
 * K; the group's check schedule), not bit for bit; the same batch always gives the same
 * results.  Utterances outside the grouped path's range (n <= 128, n >= 4096, a
 * full-spectrum request, non-cosine k-means, constraints) take the single-call path. */
int sc_predict_batch_grouped(sc_handle h, const double* const* xs, const int* ns, int d,
                             int count, const sc_config* cfg, int64_t* const* labels,
                             sc_diag* diags, int group);

/*
 * Size reduction before the spectral path (spectral_clusterer.py:170-199,
 * multi_stage_clusterer.py:109-112, fallback_clusterer.py:108-113):
 * sklearn.cluster.AgglomerativeClustering(metric="cosine", linkage=...).fit_predict(X)
 * with n_clusters > 0, or n_clusters == 0 and a distance_threshold; labels are numbered as
 * sklearn numbers them.  n_clusters_out (may be NULL) receives the cluster count.
 */
int sc_ahc(sc_handle h, const double* x, int n, int d, int linkage, int n_clusters,
           double distance_threshold, int64_t* labels, int* n_clusters_out);
/* utils.get_cluster_centroids (utils.py:159-176): out is (k, d), k = max(labels) + 1 */
int sc_cluster_centroids(sc_handle h, const double* x, int n, int d, const int64_t* labels,
                         int k, double* out);

/*
 * Fallback decisions of the callers (fallback_clusterer.py, naive_clusterer.py).
 * sc_affinity_stats: out[4] = {affinity.min(), np.diag(affinity, 1).min(), mean,
 *   np.std(affinity)} of the RESIDENT affinity (fallback_clusterer.py:137-153).
 * sc_affinity_gmm_bic: BIC of sklearn-default 1- and 2-component Gaussian mixtures fitted
 *   to the resident affinity's entries j >= i + diagonal_offset (:154-173).
 * sc_naive_cluster: NaiveClusterer.predict (naive_clusterer.py:57-105) continuing from a
 *   state of *n_centroids centroids (row-major, `capacity` rows >= *n_centroids + n).
 */
int sc_affinity_stats(sc_handle h, double* out);
int sc_affinity_gmm_bic(sc_handle h, int diagonal_offset, double* bic1, double* bic2);
int sc_naive_cluster(sc_handle h, const double* x, int n, int d, double threshold,
                     double adaptation_threshold, double* centroids, int32_t* counts,
                     int32_t* n_centroids, int capacity, int64_t* labels);

/* ---- single stages (ndarray in / ndarray out; parity tests and the
 *      per-op Python classes use these) ------------------------------------- */
/* utils.compute_affinity_matrix (utils.py:20-41) */
int sc_stage_affinity(sc_handle h, const double* x, int n, int d, double* out);
/* AffinityRefinementOperation.refine (refinement.py:136-245); op = sc_op,
 * options are read from cfg. */
int sc_stage_refine(sc_handle h, int op, const sc_config* cfg, const double* in,
                    int n, double* out);
/* ConstraintOperation.adjust_affinity (constraint.py:106-118, 138-164); the operator
 * and its options are read from cfg.  Any square affinity / constraint matrix. */
int sc_stage_constraint(sc_handle h, const sc_config* cfg, const double* affinity,
                        const double* constraint_matrix, int n, double* out);
/* rowmax / rowsum of Diffuse(a) = a a^T (refinement.py:232-234) for a SYMMETRIC (n, n) input --
 * what RowWiseNormalize (refinement.py:240-245) and the Laplacian degree (laplacian.py:41) read
 * of it -- by either route: mode 1 the explicit fp64 product, mode 2 the matrix-free search
 * (n <= 65536).  info (6 ints since ABI 7, may be NULL): candidates evaluated exactly, rows over
 * the candidate cap (evaluated in full), largest candidate count of a row, 1 if S was formed after
 * all, tiles of the digit product computed, tiles in its upper triangle. */
int sc_stage_diffuse_rowstats(sc_handle h, const double* a, int n, int mode, double* rowmax,
                              double* rowsum, int32_t* info);
/* laplacian.compute_laplacian (laplacian.py:24-60) */
int sc_stage_laplacian(sc_handle h, int laplacian_type, const double* in, int n,
                       double* out);
/*
 * utils.compute_sorted_eigenvectors (utils.py:44-71) for a SYMMETRIC input:
 * the `count` largest (descend=1) or smallest (descend=0) eigenpairs.
 * values: count doubles; vectors: (n, count), unit 2-norm columns, or NULL for values
 * only -- then count may be anything up to n (count > 64 at n > 128 takes the dense
 * tridiagonalisation path: what np.linalg.eigvalsh returns).
 */
int sc_stage_sym_eig(sc_handle h, const double* m, int n, int count, int descend,
                     double* values, double* vectors, sc_diag* diag);
/*
 * utils.compute_sorted_eigenvectors (utils.py:44-71) for ANY square input: what
 * np.linalg.eig + .real + argsort give -- real parts of the `count` eigenvalues of
 * largest (descend=1) / smallest (descend=0) real part and the real parts of their
 * eigenvectors, normalised like LAPACK dgeev (unit 2-norm, largest component of a
 * complex vector real).  n <= 64: Hessenberg + complex QR in one wavefront (all pairs
 * available); larger n: block Arnoldi, count <= 32.
 */
int sc_stage_eig(sc_handle h, const double* m, int n, int count, int descend,
                 double* values, double* vectors, sc_diag* diag);
/* numpy.random.RandomState(seed).random_sample(count): the MT19937 stream that
 * sklearn's KMeans(random_state=0) (custom_distance_kmeans.py:39-43) consumes.
 * Host-only; exported so the stream can be pinned without a GPU. */
int sc_random_state_doubles(uint32_t seed, int count, double* out);
/* index RandomState.choice(n, p=uniform) returns for the uniform draw u:
 * cumsum(1/n) / cdf[-1], searchsorted(u, side="right") (first k-means++ centre) */
int sc_uniform_choice(int n, double u);
/* The host routine that solves the small (<= 64 x 64) Rayleigh-Ritz problems of the block
 * Lanczos solver: symmetric a (m x m, row-major) -> eigenvalues ascending, eigenvectors in
 * the columns of `vectors` (Householder tridiagonalisation + implicit QL).  Host-only;
 * exported so it can be pinned without a GPU. */
int sc_host_symmetric_eig(const double* a, int m, double* values, double* vectors);
/* The Rayleigh-Ritz solve of larger bases (64 < m <= 128): ALL eigenvalues of the symmetric
 * m x m matrix `a` (row-major; upper triangle read), descending, and the eigenvectors of the
 * leading `need` of them ((m, need) row-major): Householder tridiagonalisation with kept
 * reflectors + QL for the values + inverse iteration + back-transform.  Host-only. */
int sc_host_symmetric_eig_partial(const double* a, int m, int need, double* values,
                                  double* vectors);
/* Eigenvectors of the symmetric tridiagonal matrix (d[0..n), e[0..n-1)) for the k given
 * eigenvalues `lam` by inverse iteration (LAPACK dstein's method): vectors is (n, k)
 * row-major, column q belongs to lam[q], unit 2-norm.  The host step of the dense landing
 * pad (SC_EIG_PATH_DENSE_FULL) that takes over whenever block Lanczos gives up, so that
 * predict() returns wherever np.linalg.eig (utils.py:59) does.  Host-only; exported so it
 * can be pinned without a GPU. */
int sc_host_tridiag_eigvectors(const double* d, const double* e, int n, const double* lam,
                               int k, double* vectors);
/* The Rayleigh-Ritz solve of the WIDE block Arnoldi (general eigen path, more than 32 eigenpairs
 * at n > 64: projected problems of order 64 < m <= 128; smaller ones are solved by a
 * one-wavefront device kernel): eigenvalues of the real m x m matrix `a` (row-major) sorted by
 * real part, descending, and the first nvec eigenvectors ((m, nvec) row-major, real and
 * imaginary parts, unit 2-norm).  Hessenberg reduction + shifted complex QR + back
 * substitution.  Host-only; exported so it can be pinned against numpy without a GPU. */
int sc_host_general_eig(const double* a, int m, int nvec, double* values_re, double* values_im,
                        double* vectors_re, double* vectors_im);
/* The same contract in real arithmetic -- Householder Hessenberg reduction, double-shift QR for
 * the values, inverse iteration + back-transform for the nvec leading vectors: what the
 * Rayleigh-Ritz checks of block Arnoldi (every basis size, general eigen path; replaces
 * np.linalg.eig of utils.py:59 on the projected problem) call since round 6; the function above is
 * its fallback when an inverse iteration does not converge.  Host-only; exported so it can be
 * pinned against numpy without a GPU. */
int sc_host_general_eig_fast(const double* a, int m, int nvec, double* values_re,
                             double* values_im, double* vectors_re, double* vectors_im);
/* The host half of the dense general eigensolver for n > 64 (SC_EIG_PATH_DENSE_HESSENBERG;
 * replaces np.linalg.eig, utils.py:59, where more eigenvalues of a non-symmetric matrix are read
 * than a Krylov basis holds).  `packed` (n, n) row-major: an upper Hessenberg matrix on and
 * above the subdiagonal and, below it, the Householder reflectors that produced it (LAPACK
 * dgehd2's storage, what the device reduction leaves), tau (n - 2).  values: all n eigenvalues
 * (implicit double-shift QR, unordered); vectors_re / vectors_im ((n, count) row-major): the
 * eigenvectors of the ORIGINAL matrix for the eigenvalues values[pick[q]] -- inverse iteration on
 * the Hessenberg form + back-transform through the reflectors, not normalised; *max_resid: the
 * largest relative residual on the Hessenberg form.  Host-only; exported so it can be pinned
 * against numpy without a GPU. */
int sc_host_hessenberg_eig(const double* packed, const double* tau, int n, int count,
                           const int32_t* pick, double* values_re, double* values_im,
                           double* vectors_re, double* vectors_im, double* max_resid);
/* The eigensolver's stopping rule for one Ritz value (eig_driver.hip): a bound on the distance
 * from theta[i] (Ritz values of a SYMMETRIC operator, descending) to the eigenvalue it
 * approximates, from the residual norms resid[] -- resid[i] itself, or the Kato-Temple bound
 * resid[i]^2 / delta where the neighbouring Ritz values fence theta[i] off.  Host-only;
 * exported so the bound can be checked against true Rayleigh-Ritz errors without a GPU.
 * Returns the bound through *bound. */
int sc_host_value_error_bound(const double* theta, const double* resid, int m, int i,
                              double* bound);
/* utils.compute_number_of_clusters (utils.py:74-130) -- host scalar loop */
int sc_eigengap(const double* eigenvalues, int count, int max_clusters,
                double stop_eigenvalue, int eigengap_type, int descend,
                int* n_clusters, double* max_delta);
/* custom_distance_kmeans.run_kmeans (custom_distance_kmeans.py:13-52),
 * custom_dist="cosine": sklearn k-means++ (RandomState(0)) + one Lloyd step
 * for the seeds, then the cosine loop.  e: (n, k).  centroids_out may be
 * NULL, else (k, k) final centroids. */
int sc_stage_kmeans(sc_handle h, const double* e, int n, int k, int max_iter,
                    int64_t* labels, double* centroids_out, int* iterations);
/* the same with custom_dist = SC_KMEANS_* */
int sc_stage_kmeans_metric(sc_handle h, const double* e, int n, int k, int max_iter,
                           int metric, int64_t* labels, double* centroids_out,
                           int* iterations);

/* ---- multi-GPU: replicas of the path over the GPUs of one node ------------------------
 * The reference has no distributed layer (a batch is a Python `for` over predict(),
 * SURVEY.md 3.4; AutoTune evaluates its p grid serially, autotune.py:98-111).  Independent
 * units are partitioned over GPUs by the host (spectralcluster_amd/multigpu.py); these
 * entry points are the only communication it needs, on RCCL over xGMI.  Host buffers in
 * and out (staged through the device on the handle's stream); every call returns after
 * its result is on the host.  librccl is opened on first use. */
#define SC_COMM_ID_BYTES 128
typedef struct sc_comm_s* sc_comm;
/* 1 when librccl could be opened */
int sc_comm_available(void);
/* rank 0: ncclGetUniqueId; the 128 bytes reach the other ranks out of band (file/socket) */
int sc_comm_unique_id(unsigned char* id);
/* one process per GPU: join a communicator of world_size ranks on the handle's device;
 * collectives run on the handle's stream */
int sc_comm_init_rank(sc_handle h, int world_size, int rank, const unsigned char* id,
                      sc_comm* out);
/* one process driving ndev GPUs (ncclCommInitAll): out[i] is bound to handles[i] */
int sc_comm_init_all(sc_handle* handles, int ndev, sc_comm* out);
int sc_comm_destroy(sc_comm c);
int sc_comm_rank(sc_comm c);
int sc_comm_size(sc_comm c);
const char* sc_comm_last_error(sc_comm c);
/* root's `bytes` bytes of buf reach every rank's buf (embeddings, packed sc_config) */
int sc_comm_broadcast(sc_comm c, void* buf, size_t bytes, int root);
/* recv (world_size * bytes) = every rank's send (bytes), in rank order (label slabs,
 * AutoTune (ratio, n_clusters) pairs) */
int sc_comm_allgather(sc_comm c, const void* send, void* recv, size_t bytes);
/* element-wise max over ranks, in place (the bench's max-over-ranks time) */
int sc_comm_allreduce_max(sc_comm c, double* values, int count);
/* every rank's stream has drained and every rank has arrived */
int sc_comm_barrier(sc_comm c);

#ifdef __cplusplus
}
#endif
#endif /* SPECTRALCLUSTER_AMD_H_ */
