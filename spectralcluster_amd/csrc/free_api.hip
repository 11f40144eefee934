// Host side of the matrix-free Diffuse (kernels: diffuse_free.hip): when the
// route is taken, the statistics pipeline, the exact evaluation of rows the candidate search
// cannot prune, and the two-pass operator the eigensolver applies instead of reading S.
#include "handle.h"

namespace {
constexpr int kOvfWords = 80;   // [0] rows over the cap, [1..64] their ids, [65] candidates, [66] max,
                                // [67] tiles of the digit product that ran, [68] length of its skip list
constexpr int kOvfRowsMax = 64;
}  // namespace

extern "C" int sc_set_diffuse_mode(sc_handle h, int mode) {
  if (!h || mode < -1 || mode > 2) return SC_ERR_INVALID;
  h->diffuse_mode = mode;
  return SC_OK;
}

extern "C" int sc_set_free_prune(sc_handle h, int on) {
  if (!h || on < -1 || on > 1) return SC_ERR_INVALID;
  h->free_prune = on;
  return SC_OK;
}
static bool free_prune_on(sc_handle h) {
  return h->free_prune >= 0 ? h->free_prune != 0 : !sw::free_no_prune();
}

bool free_diffuse_wanted(sc_handle h, const sc_config* cfg, int n, const EigRequest& rq,
                         bool in_group) {
  // the call's own choice, else the handle's, else the process default
  const int mode = cfg->diffuse_mode == 1 || cfg->diffuse_mode == 2
                       ? cfg->diffuse_mode
                       : (h->diffuse_mode >= 0 ? h->diffuse_mode : sw::diffuse_mode());
  if (mode == 1) return false;
  // the dense routes read entries of the operator; n <= 128 is one Jacobi launch on it; the
  // i32 accumulators of the digit products hold K <= 65536
  if (n <= kDenseMax || n > 65536 || wants_full_spectrum(rq)) return false;
  if (mode == 2) return true;
  // (a single call is a chain of launch latencies up to n ~ 2000, where the route's extra
  //  launches cost what its smaller product saves: 0.80 / 0.81 ms at n = 1792, 0.90 / 0.87 at
  //  2048; in a group the latencies are shared and the work decides: 6044-6315 utterances/s
  //  on config 5 with the members from 1536 on, 5782-5913 from 2048 on -- profiles/r06e)
  return n >= (in_group ? sw::diffuse_free_min_n_group() : sw::diffuse_free_min_n());
}

int ensure_free(sc_handle h, int n) {
  const size_t nv = (size_t)round_up(n, 16) * sizeof(double);
  SC_TRY(grow(h, h->fq, free_q_bytes(n)));
  SC_TRY(grow(h, h->ft32, free_t32_bytes(n)));
  SC_TRY(grow(h, h->fy1, nv));
  SC_TRY(grow(h, h->fR, nv));
  SC_TRY(grow(h, h->fscal, 4 * sizeof(double)));
  SC_TRY(grow(h, h->fwords, ((size_t)2 * n + kOvfWords) * sizeof(int)));
  SC_TRY(grow(h, h->fcand, (size_t)n * free_candidate_cap() * sizeof(int)));
  SC_TRY(grow(h, h->fY, (size_t)n * kEigBlock * sizeof(double)));
  SC_TRY(grow(h, h->fq2part, free_q2part_bytes(n)));
  SC_TRY(grow(h, h->fmx64, free_mx64_bytes(n)));
  SC_TRY(grow(h, h->ftau64, free_tau64_bytes(n)));
  SC_TRY(grow(h, h->fplan, free_plan_bytes(n)));
  if (!h->h_free)
    SC_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_free), kOvfWords * sizeof(int)));
  return SC_OK;
}

static FreeSegs free_segs(sc_handle h) {
  return FreeSegs{ptr<double>(h->fq2part), ptr<double>(h->fmx64), ptr<float>(h->ftau64)};
}

// the integer product of the handle's digits over the tiles of its skip list (the pass that wrote
// the digits left mx64 / tau64; split-K tail on the handle's workspace)
int free_product(sc_handle h, hipStream_t s, int n) {
  // (the workspace of the split-K tail belongs to handles that launch a product of their own:
  //  the members of a sweep share one grouped launch and never need it)
  SC_TRY(grow(h, h->fsplit, free_i8_split_bytes_plan()));
  launch_free_tile_flags(s, ptr<double>(h->fmx64), ptr<float>(h->ftau64), n, ptr<int>(h->fplan),
                         free_prune_on(h), ptr<int>(h->fwords));
  launch_gemm_i8_sym(s, ptr<signed char>(h->fq), n, h->tilemap_cur, ptr<float>(h->ft32),
                     ptr<unsigned>(h->fwords), ptr<int>(h->fsplit), ptr<int>(h->fplan));
  return SC_OK;
}

// the pipeline in three pieces, all on stream `s` (the handle's own for a single call; the
// sweep's for a member arena, whose product is one grouped launch for all members)
int free_stats_begin(sc_handle h, hipStream_t s, const double* A, int ld, int n, bool have_amax) {
  SC_TRY(ensure_free(h, n));
  // ([0] max|a| stays when the caller has it; [2] max R starts from zero)
  SC_HIP(h, hipMemsetAsync(ptr<double>(h->fscal) + (have_amax ? 1 : 0), 0,
                           (have_amax ? 3 : 4) * sizeof(double), s));
  SC_HIP(h, hipMemsetAsync(h->fwords.p, 0, ((size_t)2 * n + kOvfWords) * sizeof(int), s));
  if (!have_amax) launch_free_absmax(s, A, n, ld, ptr<double>(h->fscal));
  launch_free_quantize(s, A, n, ld, ptr<signed char>(h->fq), ptr<double>(h->fscal),
                       ptr<double>(h->fy1), ptr<double>(h->fR), ptr<double>(h->fq2part));
  launch_free_seg_reduce(s, ptr<double>(h->fR), n, free_segs(h));
  return SC_OK;
}
int free_stats_end(sc_handle h, hipStream_t s, const double* A, int ld, int n, bool timed,
                   const int* plan) {
  unsigned* M = ptr<unsigned>(h->fwords);
  int* count = ptr<int>(h->fwords) + n;
  int* ovf = ptr<int>(h->fwords) + 2 * (size_t)n;
  launch_t32_candidates(s, ptr<float>(h->ft32), n, M, ptr<double>(h->fR), ptr<double>(h->fscal),
                        count, ptr<int>(h->fcand), plan);
  if (timed) ev_rec(h, &h->free_ev[3]);
  launch_free_row_stats(s, A, n, ld, ptr<double>(h->fy1), count, ptr<int>(h->fcand),
                        ptr<double>(h->rowmax), ptr<double>(h->rowsum), ovf);
  if (timed) ev_rec(h, &h->free_ev[4]);
  SC_TRY(check_last(h, "matrix-free diffuse launch"));
  SC_HIP(h, hipMemcpyAsync(h->h_free, ovf, kOvfWords * sizeof(int), hipMemcpyDeviceToHost, s));
  h->free_checked = false;
  return SC_OK;
}

// The same two pieces for a GROUP of matrices of ICASSP fronts (AutoTune sweep: 16 members of
// one size): the begin step, quantiser, candidate scan and exact statistics of all members are
// one launch each -- a member's own kernels of 20-100 us left most of the chip idle between
// their tails, and 16 x 7 launches were 16 x 7 launch latencies.  The caller puts the digit
// product (one grouped launch, or one per member) between the two.
int free_group_begin(sc_handle* hs, const double* const* A, const double* const* cuts,
                     const double* ps, int count, const int* lds, const int* ns, hipStream_t s,
                     double floor_value, FreeItem* items) {
  for (int z = 0; z < count; ++z) {
    sc_handle h = hs[z];
    const int n = ns[z], ld = lds[z];
    SC_TRY(ensure_free(h, n));
    items[z] = FreeItem{A[z], n, ld, ptr<signed char>(h->fq), ptr<float>(h->ft32),
                        ptr<int>(h->fwords), ptr<double>(h->fscal), ptr<double>(h->fy1),
                        ptr<double>(h->fR), ptr<int>(h->fcand), ptr<double>(h->rowmax),
                        ptr<double>(h->rowsum), cuts[z], ps[z]};
    items[z].q2part = ptr<double>(h->fq2part);
    items[z].mx64 = ptr<double>(h->fmx64);
    items[z].tau64 = ptr<float>(h->ftau64);
    items[z].plan = ptr<int>(h->fplan);
  }
  launch_free_begin_group(s, items, count, floor_value);
  launch_free_quantize_group(s, items, count);
  // the members' skip lists (the caller hands items[z].plan to the grouped product)
  launch_free_tile_flags_group(s, items, count, free_prune_on(hs[0]), true);
  return SC_OK;
}
// Round 6: the grouped threshold + symmetrise pass writes the members' digits itself (what the
// single call has done since round 4) -- the quantiser's extra read of every member's matrix goes
// (k_free_quantize_g: 0.57 ms of a 16-value sweep at n = 4096, 3.3 % of config 5's GPU time).
//   free_group_prepare   BEFORE the threshold pass, once the cut vectors are there: buffers, max|a|
//                        from the cuts, words cleared, the digit rows no tile writes cleared; fills
//                        items[] and digits[] (what launch_threshold_symmetrize_group takes)
//   free_group_digits    AFTER it: y1 / R / max R / the skip-list thresholds from the row partials,
//                        then the members' skip lists
// then the grouped product and free_group_end as before.  SC_GROUP_QUANTIZE_PASS=1: rounds 4-5's
// separate quantiser (free_group_begin).
int free_group_prepare(sc_handle* hs, const double* const* A, const double* const* cuts,
                       const double* ps, int count, const int* lds, const int* ns, hipStream_t s,
                       double floor_value, FreeItem* items, TsDigits* digits) {
  for (int z = 0; z < count; ++z) {
    sc_handle h = hs[z];
    const int n = ns[z], ld = lds[z];
    SC_TRY(ensure_free(h, n));
    const size_t rows64 = (size_t)round_up(n, 64), nblk = rows64 / 64;
    SC_TRY(grow(h, h->fypart, rows64 * nblk * sizeof(double)));
    SC_TRY(grow(h, h->frpart, rows64 * nblk * sizeof(int)));
    items[z] = FreeItem{A[z], n, ld, ptr<signed char>(h->fq), ptr<float>(h->ft32),
                        ptr<int>(h->fwords), ptr<double>(h->fscal), ptr<double>(h->fy1),
                        ptr<double>(h->fR), ptr<int>(h->fcand), ptr<double>(h->rowmax),
                        ptr<double>(h->rowsum), cuts[z], ps[z]};
    items[z].q2part = ptr<double>(h->fq2part);
    items[z].mx64 = ptr<double>(h->fmx64);
    items[z].tau64 = ptr<float>(h->ftau64);
    items[z].plan = ptr<int>(h->fplan);
    items[z].ypart = ptr<double>(h->fypart);
    items[z].rpart = ptr<int>(h->frpart);
    digits[z] = TsDigits{ptr<signed char>(h->fq), (size_t)2 * rows64, (int)nblk,
                         ptr<double>(h->fscal), ptr<double>(h->fypart), ptr<int>(h->frpart),
                         ptr<double>(h->fq2part), ptr<double>(h->fmx64)};
  }
  launch_free_begin_group(s, items, count, floor_value, true);
  return SC_OK;
}
int free_group_digits(sc_handle* hs, const FreeItem* items, int count, hipStream_t s) {
  launch_free_partials_reduce_group(s, items, count);
  launch_free_tile_flags_group(s, items, count, free_prune_on(hs[0]), false);
  return SC_OK;
}

int free_group_end(sc_handle* hs, const FreeItem* items, int count, hipStream_t s) {
  launch_free_scan_stats_group(s, items, count);
  SC_TRY(check_last(hs[0], "matrix-free diffuse launch"));
  for (int z = 0; z < count; ++z) {
    sc_handle h = hs[z];
    SC_HIP(h, hipMemcpyAsync(h->h_free, items[z].words + 2 * (size_t)items[z].n,
                             kOvfWords * sizeof(int), hipMemcpyDeviceToHost, s));
    h->free_checked = false;
  }
  return SC_OK;
}

// The threshold + symmetrise pass can write the digits itself (rowops.hip, TsDigits): what it
// needs before it runs -- the buffers, zeroed words, max|a| from the cut vector in fscal[0] ...
int free_fused_prepare(sc_handle h, hipStream_t s, int n, const double* cut, double p,
                       double floor_value) {
  SC_TRY(ensure_free(h, n));
  const size_t rows64 = (size_t)round_up(n, 64), nblk = rows64 / 64;
  SC_TRY(grow(h, h->fypart, rows64 * nblk * sizeof(double)));
  SC_TRY(grow(h, h->frpart, rows64 * nblk * sizeof(int)));
  // max|a| from the cut vector, the other scalars and the words (M | count | ovf) cleared: ONE
  // launch -- the group form with one member -- where rounds 4-5 had two fills and a reduction,
  // each a dependent operation of ~4 us on the stream
  FreeItem item{};
  item.n = n;
  item.words = ptr<int>(h->fwords);
  item.scal = ptr<double>(h->fscal);
  item.cut = cut;
  item.p = p;
  launch_free_begin_group(s, &item, 1, floor_value);
  return SC_OK;
}

// (digits_ready: ... and what is left of the quantiser's work afterwards, the row partials)
int free_diffuse_stats(sc_handle h, const double* A, int ld, int n, bool have_amax,
                       bool digits_ready) {
  hipStream_t s = h->stream;
  SC_TRY(ensure_tilemap(h, n));
  ev_rec(h, &h->free_ev[0]);
  if (digits_ready)
    launch_free_partials_reduce(s, ptr<double>(h->fypart), ptr<int>(h->frpart), n,
                                ptr<double>(h->fy1), ptr<double>(h->fR), ptr<double>(h->fscal),
                                free_segs(h));
  else
    SC_TRY(free_stats_begin(h, s, A, ld, n, have_amax));
  ev_rec(h, &h->free_ev[1]);
  SC_TRY(free_product(h, s, n));
  ev_rec(h, &h->free_ev[2]);
  return free_stats_end(h, s, A, ld, n, true, ptr<int>(h->fplan));
}

// plain product W = A Vs by the solver's own block matvec (c = 1, p = 0)
static void plain_matvec(sc_handle h, const double* A, int ld, int n, bool sym_mv,
                         const double* Vs, double* W) {
  if (sym_mv)
    launch_block_matvec_sym(h->stream, A, ld, n, nullptr, nullptr, Vs, kEigBlock, Vs, W,
                            ptr<double>(h->mvsym));
  else
    launch_block_matvec(h->stream, A, ld, n, nullptr, nullptr, Vs, kEigBlock, Vs, W);
}

int free_fix_overflow(sc_handle h, const double* A, int ld, int n, bool* changed,
                      bool* too_many) {
  *changed = *too_many = false;
  hipStream_t s = h->stream;
  SC_HIP(h, hipStreamSynchronize(s));  // (h_free was copied behind the statistics)
  h->free_checked = true;
  const int rows = h->h_free[0];
  if (rows <= 0) return SC_OK;
  if (rows > kOvfRowsMax) {
    *too_many = true;
    return SC_OK;
  }
  // S[:, rows] = A (A[rows, :]^T), eight rows per pass: two block products over A
  SC_TRY(ensure_eig(h, n));
  const bool sym_mv = n >= sw::matvec_sym_min_n();
  const int* ids = ptr<int>(h->fwords) + 2 * (size_t)n + 1;
  for (int r0 = 0; r0 < rows; r0 += kEigBlock) {
    const int nr = std::min(kEigBlock, rows - r0);
    launch_free_gather_rows(s, A, n, ld, ids + r0, nr, ptr<double>(h->Vs));
    plain_matvec(h, A, ld, n, sym_mv, ptr<double>(h->Vs), ptr<double>(h->W));
    launch_free_colmax(s, ptr<double>(h->W), n, ids + r0, nr, ptr<double>(h->rowmax));
  }
  SC_TRY(check_last(h, "matrix-free diffuse: exact rows"));
  *changed = true;
  return SC_OK;
}

void free_apply_operator(sc_handle h, const double* A, int ld, int n, bool sym_mv,
                         const double* V, int ldv) {
  // fY = A Vs (Vs = c .* V, left by the chain), then W = p .* V + c .* (A fY)
  plain_matvec(h, A, ld, n, sym_mv, ptr<double>(h->Vs), ptr<double>(h->fY));
  if (sym_mv)
    launch_block_matvec_sym(h->stream, A, ld, n, ptr<double>(h->cvec), ptr<double>(h->pvec), V,
                            ldv, ptr<double>(h->fY), ptr<double>(h->W), ptr<double>(h->mvsym));
  else
    launch_block_matvec(h->stream, A, ld, n, ptr<double>(h->cvec), ptr<double>(h->pvec), V, ldv,
                        ptr<double>(h->fY), ptr<double>(h->W));
}

// rowmax(S), rowsum(S) of S = a a^T for a symmetric (n, n) input by either route (parity
// tests): mode 1 the fp64 MFMA product with fused row statistics, mode 2 the matrix-free search.
// info (may be NULL): [0] candidates evaluated, [1] rows over the cap, [2] largest candidate
// count of a row, [3] 1 when the exact-row route gave up and S was formed after all.
extern "C" int sc_stage_diffuse_rowstats(sc_handle h, const double* a, int n, int mode,
                                         double* rowmax, double* rowsum, int32_t* info) {
  if (!h) return SC_ERR_INVALID;
  if (!a || n <= 0 || !rowmax || !rowsum || (mode != 1 && mode != 2))
    return fail(h, SC_ERR_INVALID, "bad diffuse row statistics request");
  if (mode == 2 && n > 65536) return fail(h, SC_ERR_UNSUPPORTED, "matrix-free diffuse: n <= 65536");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, n, 0));
  SC_TRY(ensure_eig(h, n));
  SC_TRY(ensure_tilemap(h, n));
  const int ld = matrix_ld(n);
  h->n = n;
  h->ldn = ld;
  h->have_affinity = h->have_cropval = false;
  h->have_x = false;
  h->n_vec = 0;
  h->nev = 0;
  hipStream_t s = h->stream;
  double* A = ptr<double>(h->B2);
  SC_TRY(h2d_matrix(h, a, n, n, A, ld));
  int inf[6] = {0, 0, 0, 0, 0, 0};
  auto explicit_stats = [&]() -> int {
    GemmRowStats rs{1, ptr<double>(h->statp), ptr<double>(h->statp) + (size_t)n * gemm_tile_dim(n),
                    ptr<double>(h->rowmax), ptr<double>(h->rowsum)};
    launch_gemm_nt(s, A, ld, A, ld, ptr<double>(h->B1), ld, n, n, n, kEpiNone, true,
                   ptr<double>(h->splitk), h->tilemap_cur, &rs);
    return check_last(h, "diffuse launch");
  };
  if (mode == 1) {
    SC_TRY(explicit_stats());
  } else {
    SC_TRY(free_diffuse_stats(h, A, ld, n));
    bool changed = false, too_many = false;
    SC_TRY(free_fix_overflow(h, A, ld, n, &changed, &too_many));
    inf[0] = h->h_free[65];
    inf[1] = h->h_free[0];
    inf[2] = h->h_free[66];
    inf[4] = h->h_free[67];
    inf[5] = gemm_tile_dim(n) * (gemm_tile_dim(n) + 1) / 2;
    if (too_many) {
      inf[3] = 1;
      SC_TRY(explicit_stats());
    }
  }
  SC_HIP(h, hipMemcpyAsync(rowmax, h->rowmax.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(rowsum, h->rowsum.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  if (info) memcpy(info, inf, sizeof(inf));
  return SC_OK;
}
