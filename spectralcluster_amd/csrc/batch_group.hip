// Grouped execution of a batch of SHORT utterances (512 independent
// predict() calls, n = 300..3000): groups of up to 16 utterances per launch, dealt to three
// lanes (a lead handle, its streams and a host thread each; sc_predict_batch_grouped below).
//
// A short utterance cannot fill 256 CUs, and after its three GEMM-shaped stages its pipeline
// is ~50 tiny dependent launches (block Lanczos chain, k-means chain) with three host
// synchronisations: 0.45-0.6 ms of latency that does not shrink with n.  Running utterances
// on several streams from several host threads hides that latency (sc_predict_batch_streams);
// this file removes it instead.  Per group:
//   front   upload, affinity GEMM, refinement, Diffuse GEMM, scaling vectors of every member
//           of a group of up to kGroupMax utterances are enqueued back to back, each on the
//           member's own stream (no host synchronisation; one utterance's GEMM tiles cannot
//           fill the chip, several members' do); an event per member hands over to the
//           owner's stream,
//   eigen   ONE lockstep block Lanczos for the group (eig_driver.hip: sym_topk_group): every
//           launch carries one link of every member (blockIdx.y = member, descriptors in the
//           kernel arguments), one synchronisation per Rayleigh-Ritz check for all of them,
//   k-means ONE lockstep chain for the group (kmeans_chain.hip: launch_kmeans_chain_group).
// (When the refinement sequence is the ICASSP2018 one, the front itself is grouped launches
// too -- enqueue_front_grouped: both GEMMs take the tiles of all members in one launch.)
// Each member runs the same kernel bodies with the same arguments as a single call; members
// that leave the common path (rare branches of the eigensolver, k > 32, a non-symmetric
// refinement, n <= 128 or n >= 4096) go through the single-call path.  Results agree with
// sc_predict to the solver's tolerance (the grouped GEMMs sum whole K tiles, the group sets
// the check schedule), and a batch call is a deterministic function of its input.
#include <ctime>
#include <thread>

#include "handle.h"

namespace {

double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

constexpr int kRndStride = 256;  // doubles per k in the RandomState(0) table (k <= 32)

int group_slot(sc_handle lead, int z, sc_handle* out) {
  while ((int)lead->gslots.size() <= z) {
    sc_handle sub = nullptr;
    const int rc = sc_create(lead->device, &sub);
    if (rc != SC_OK) return fail(lead, rc, "could not create a member arena for the group");
    // the member's stages before the eigensolver run on its own stream (several members'
    // GEMMs and refinement passes share the chip); `sync_ev` hands the result to the
    // owner's stream, where the lockstep chains run
    if (hipEventCreateWithFlags(&sub->sync_ev, hipEventDisableTiming) != hipSuccess) {
      sc_destroy(sub);
      return fail(lead, SC_ERR_HIP, "could not create a member event");
    }
    sub->profile_level = 1;
    lead->gslots.push_back(sub);
  }
  *out = lead->gslots[z];
  // (weights of a blur radius above 32 live in the handle: members inherit the lead's)
  if ((*out)->blur_ext.size() != lead->blur_ext.size() || !lead->blur_ext.empty())
    (*out)->blur_ext = lead->blur_ext;
  return SC_OK;
}

int ensure_seed_table(sc_handle lead) {
  if (lead->gkrnd_ready) return SC_OK;
  std::vector<double> table((size_t)33 * kRndStride, 0.0), rnd;
  for (int k = 1; k <= 32; ++k) {
    double u;
    int trials;
    kmeans_seed_constants(k, &u, &trials, &rnd);
    if (rnd.size() > (size_t)kRndStride) return fail(lead, SC_ERR_UNSUPPORTED, "seed table");
    std::copy(rnd.begin(), rnd.end(), table.begin() + (size_t)k * kRndStride);
  }
  SC_TRY(grow(lead, lead->gkrnd, table.size() * sizeof(double)));
  SC_HIP(lead, hipMemcpyAsync(lead->gkrnd.p, table.data(), table.size() * sizeof(double),
                              hipMemcpyHostToDevice, lead->stream));
  SC_HIP(lead, hipStreamSynchronize(lead->stream));  // `table` is a local
  lead->gkrnd_ready = true;
  return SC_OK;
}

// sc_set_embeddings without its synchronisation: the caller's arrays stay valid for the whole
// batch call
int upload_embeddings(sc_handle h, const double* x, int n, int d) {
  if (!x || n <= 0 || d <= 0) return fail(h, SC_ERR_INVALID, "embeddings must be (n, d)");
  SC_TRY(ensure_matrices(h, n, d));
  h->n = n;
  h->d = d;
  h->ldn = matrix_ld(n);
  h->ldx = round_up(d, 16);
  h->have_affinity = h->have_cropval = false;
  h->n_vec = 0;
  SC_HIP(h, hipMemcpy2DAsync(h->X.p, (size_t)h->ldx * sizeof(double), x,
                             (size_t)d * sizeof(double), (size_t)d * sizeof(double), n,
                             hipMemcpyHostToDevice, h->stream));
  h->have_x = true;
  return SC_OK;
}

struct Member {
  int index = -1;        // utterance
  sc_handle h = nullptr;
  FrontResult front;
  int state = 0;         // 0 in the group, 1 single-call path, 2 done
  int k = 0;
  bool free_op = false;  // matrix-free Diffuse: front.matrix is A, the solver applies it twice
};

// Stages before the eigensolver of one group, member after member, each on its member's
// stream: no synchronisation.  `slot0`: first member arena of the bank the group uses.
int enqueue_front(sc_handle lead, const double* const* xs, const int* ns, int d,
                  const sc_config* cfg, sc_diag* diags, const int* idx, int count, int slot0,
                  Member* mb) {
  for (int z = 0; z < count; ++z) {
    Member& m = mb[z];
    m = Member();
    m.index = idx[z];
    SC_TRY(group_slot(lead, slot0 + z, &m.h));
    sc_handle h = m.h;
    h->err.clear();
    int rc = upload_embeddings(h, xs[m.index], ns[m.index], d);
    h->nev = 0;
    sc_diag local;
    sc_diag* dg = diags ? diags + m.index : &local;
    memset(dg, 0, sizeof(*dg));
    if (rc == SC_OK) rc = sc_compute_affinity(h);
    if (rc == SC_OK) rc = eig_ncluster_impl(h, cfg, dg, &m.front);
    if (rc != SC_OK) {
      lead->err = h->err;
      return rc;
    }
    if (!m.front.symmetric) m.state = 1;
    SC_HIP(lead, hipEventRecord(h->sync_ev, h->stream));
  }
  return SC_OK;
}

// The ICASSP2018 sequence (CropDiagonal, GaussianBlur, RowWiseThreshold, Symmetrize, Diffuse,
// RowWiseNormalize with the fusions eig_ncluster_impl applies to it: crop value out of the
// affinity epilogue, blur with the diagonal override and per-strip row maxima, threshold +
// symmetrise in one pass, row statistics out of the Diffuse epilogue, RowWiseNormalize folded
// into the scaling vectors) is what the grouped front covers; anything else takes the
// member-by-member front above.
bool grouped_front_covers(const sc_config* cfg) {
  static const int seq[6] = {SC_OP_CROP_DIAGONAL, SC_OP_GAUSSIAN_BLUR, SC_OP_ROW_WISE_THRESHOLD,
                             SC_OP_SYMMETRIZE, SC_OP_DIFFUSE, SC_OP_ROW_WISE_NORMALIZE};
  if (cfg->n_ops != 6) return false;
  for (int i = 0; i < 6; ++i)
    if (cfg->ops[i] != seq[i]) return false;
  return (cfg->blur_radius == 4 || cfg->blur_radius == 8) &&
         cfg->threshold_type == SC_THRESHOLD_ROW_MAX && !cfg->preserve_diagonal;
}

// Stages before the eigensolver of one group as GROUPED launches on the bank's stream: the
// two GEMMs take the tiles of all members in one launch each (one utterance's 36..300 tiles
// cannot fill the chip, those of 16 can), the passes between them one launch each with
// blockIdx.y / z = member.  Same kernel bodies and arguments per member as
// sc_compute_affinity + eig_ncluster_impl; the GEMMs compute every tile whole (a single call
// splits the tiles of a short utterance over K: the sums differ in the last bits).
int enqueue_front_grouped(sc_handle lead, const double* const* xs, const int* ns, int d,
                          const sc_config* cfg, sc_diag* diags, const int* idx, int count,
                          int slot0, Member* mb, int bank) {
  hipStream_t s = lead->gbank_stream[bank];
  FrontItem fi[kGroupMax];
  GemmGroupItem aff[kGroupMax], dif[kGroupMax];
  memset(fi, 0, sizeof(fi));
  for (int z = 0; z < count; ++z) {
    Member& m = mb[z];
    m = Member();
    m.index = idx[z];
    SC_TRY(group_slot(lead, slot0 + z, &m.h));
    sc_handle h = m.h;
    h->err.clear();
    const int n = ns[m.index];
    if (!xs[m.index] || n <= 0 || d <= 0) return fail(lead, SC_ERR_INVALID, "embeddings must be (n, d)");
    int rc = ensure_matrices(h, n, d);
    if (rc == SC_OK) rc = ensure_eig(h, n);
    if (rc == SC_OK) rc = ensure_tilemap(h, n);
    if (rc == SC_OK) rc = grow(h, h->symflag, 16);
    if (rc != SC_OK) {
      lead->err = h->err;
      return rc;
    }
    h->n = n;
    h->d = d;
    h->ldn = matrix_ld(n);
    h->ldx = round_up(d, 16);
    h->n_vec = 0;
    h->nev = 0;
    h->have_x = h->have_affinity = h->have_cropval = true;
    h->affinity_symmetric = h->affinity_from_embeddings = true;
    h->constraint_applied = false;
    if (diags) memset(diags + m.index, 0, sizeof(sc_diag));
    SC_HIP(lead, hipMemcpy2DAsync(h->X.p, (size_t)h->ldx * sizeof(double), xs[m.index],
                                  (size_t)d * sizeof(double), (size_t)d * sizeof(double), n,
                                  hipMemcpyHostToDevice, s));
    FrontItem& f = fi[z];
    f.X = ptr<double>(h->X);
    f.Xn = ptr<double>(h->Xn);
    f.ldx = h->ldx;
    f.n = n;
    f.d = d;
    f.ldn = h->ldn;
    f.A0 = ptr<double>(h->A0);
    f.B1 = ptr<double>(h->B1);
    f.B2 = ptr<double>(h->B2);
    f.cropval = ptr<double>(h->cropval);
    f.rmpart = ptr<double>(h->rmpart);
    f.blur_cols = blur_stream_columns(n, cfg->blur_radius);  // (the grouped front streams)
    f.cut = ptr<double>(h->cut);
    f.rowmax = ptr<double>(h->rowmax);
    f.rowsum = ptr<double>(h->rowsum);
    f.cvec = ptr<double>(h->cvec);
    f.pvec = ptr<double>(h->pvec);
    f.tvec = ptr<double>(h->tvec);
    f.symflag = ptr<int>(h->symflag);
    f.flags = ptr<int>(h->flags);
    const int nt = gemm_tile_dim(n);
    aff[z] = GemmGroupItem();
    aff[z].A = f.Xn;
    aff[z].lda = h->ldx;
    aff[z].C = f.A0;
    aff[z].ldc = h->ldn;
    aff[z].n = n;
    aff[z].K = d;
    aff[z].tilemap = h->tilemap_cur;
    aff[z].partial_max = ptr<double>(h->statp);
    aff[z].rowmax = ptr<double>(h->cropval);
    dif[z] = GemmGroupItem();
    dif[z].A = f.B2;
    dif[z].lda = h->ldn;
    dif[z].C = f.B1;
    dif[z].ldc = h->ldn;
    dif[z].n = n;
    dif[z].K = n;
    dif[z].tilemap = h->tilemap_cur;
    dif[z].partial_max = ptr<double>(h->statp);
    dif[z].partial_sum = ptr<double>(h->statp) + (size_t)n * nt;
    dif[z].rowmax = ptr<double>(h->rowmax);
    dif[z].rowsum = ptr<double>(h->rowsum);
    m.front.matrix = f.B1;
    m.front.scratch = f.B2;
    m.front.ld = h->ldn;
    m.front.symmetric = true;
    m.front.folded_rownorm = true;
    // Large members take the matrix-free Diffuse (free_api.hip): their n^3 product is most of
    // a batch's GEMM time (78 % of config 5's Diffuse flops sit in utterances of n >= 2048;
    // members switch from n = 1536 on: free_diffuse_wanted)
    m.free_op = free_diffuse_wanted(lead, cfg, n, make_eig_request(cfg), true) &&
                cfg->soft_multiplier >= 0.0 && cfg->soft_multiplier <= 1.0 &&
                cfg->p_percentile > 0.0;
    if (m.free_op) {
      rc = ensure_free(h, n);
      if (rc != SC_OK) {
        lead->err = h->err;
        return rc;
      }
      dif[z].n = 0;          // idle in the grouped fp64 product
      m.front.matrix = f.B2;  // the thresholded + symmetrised A stays the operand
      m.front.scratch = f.B1;
      m.front.free_op = true;  // (a hand-back resumes on the two-pass operator: api.hip)
    }
  }
  launch_front_begin_group(s, fi, count, true);
  launch_gemm_nt_group(s, aff, count, kEpiAffinity, 2);
  launch_gaussian_blur_group(s, fi, count, cfg->blur_radius, ptr<double>(lead->blurw));
  // rowmax / rowsum of S = A A^T without forming it, for the members that take that route (the
  // cut vector bounds max|a|: the grouped front is the ICASSP2018 sequence on a cosine affinity):
  // begin, scan and statistics are one launch each for all of them, the digit products one
  // launch with a member per blockIdx.y; round 6: their digits come out of the threshold pass
  sc_handle fh[kGroupMax];
  const double* mats[kGroupMax];
  const double* cuts[kGroupMax];
  double ps[kGroupMax];
  int nn[kGroupMax], ll[kGroupMax], nf = 0, fz[kGroupMax];
  FreeItem fitems[kGroupMax];
  for (int z = 0; z < count; ++z) {
    if (!mb[z].free_op) continue;
    sc_handle h = mb[z].h;
    fh[nf] = h;
    fz[nf] = z;
    mats[nf] = fi[z].B2;
    cuts[nf] = ptr<double>(h->cut);
    ps[nf] = cfg->p_percentile;
    nn[nf] = h->n;
    ll[nf] = h->ldn;
    ++nf;
  }
  const double amax_floor = (cfg->binarize || cfg->preserve_diagonal) ? 1.0 : 0.0;
  const bool fused_digits = nf > 0 && !sw::group_quantize_pass();
  if (fused_digits) {
    // (the cuts first -- max|a| comes from them --, then the begin step, then the pass itself)
    launch_cut_from_partials_group(s, fi, count, cfg->p_percentile);
    TsDigits packed[kGroupMax], digits[kGroupMax];
    memset(digits, 0, sizeof(digits));
    SC_TRY(free_group_prepare(fh, mats, cuts, ps, nf, ll, nn, s, amax_floor, fitems, packed));
    for (int q = 0; q < nf; ++q) digits[fz[q]] = packed[q];
    launch_threshold_symmetrize_group(s, fi, count, cfg->p_percentile, cfg->soft_multiplier,
                                      cfg->binarize, cfg->symmetrize_type, cfg->preserve_diagonal,
                                      true, digits);
  } else {
    launch_threshold_symmetrize_group(s, fi, count, cfg->p_percentile, cfg->soft_multiplier,
                                      cfg->binarize, cfg->symmetrize_type, cfg->preserve_diagonal);
  }
  launch_gemm_nt_group(s, dif, count, kEpiNone, 1);
  {
    if (nf > 0) {
      if (fused_digits)
        SC_TRY(free_group_digits(fh, fitems, nf, s));
      else
        SC_TRY(free_group_begin(fh, mats, cuts, ps, nf, ll, nn, s, amax_floor, fitems));
      {  // the digit products of all of them: one launch (blockIdx.y = member)
        const signed char* qs[kGroupMax];
        float* ts[kGroupMax];
        unsigned* ms[kGroupMax];
        const int2* tms[kGroupMax];
        const int* pls[kGroupMax];
        for (int q = 0; q < nf; ++q) {
          qs[q] = ptr<signed char>(fh[q]->fq);
          ts[q] = ptr<float>(fh[q]->ft32);
          ms[q] = ptr<unsigned>(fh[q]->fwords);
          tms[q] = fh[q]->tilemap_cur;
          pls[q] = fitems[q].plan;
        }
        launch_gemm_i8_sym_group(s, qs, ts, ms, nf, nn, tms, pls);
      }
      SC_TRY(free_group_end(fh, fitems, nf, s));
    }
  }
  launch_scaling_vectors_group(s, fi, count, cfg->laplacian_type, 1);
  SC_TRY(check_last(lead, "grouped front launch"));
  SC_HIP(lead, hipEventRecord(lead->gbank_ev[bank], s));
  return SC_OK;
}

// Eigensolver and k-means of a group whose stages before are enqueued (enqueue_front), in
// lockstep on the owner's stream; then the members that left the common path.
int finish_group(sc_handle lead, const int* ns, const sc_config* cfg, int64_t* const* labels,
                 sc_diag* diags, Member* mb, int count, const EigRequest& rq, int front_bank) {
  hipStream_t s = lead->stream;
  const bool trace = sw::group_trace();
  if (front_bank >= 0) {
    SC_HIP(lead, hipStreamWaitEvent(s, lead->gbank_ev[front_bank], 0));
  } else {
    for (int z = 0; z < count; ++z) SC_HIP(lead, hipStreamWaitEvent(s, mb[z].h->sync_ev, 0));
  }
  const double t1 = trace ? now_us() : 0.0;
  // ---- eigen: lockstep over the symmetric members
  GroupEigMember em[kGroupMax];
  int emz[kGroupMax], ne = 0;
  for (int z = 0; z < count; ++z) {
    if (mb[z].state != 0) continue;
    em[ne].h = mb[z].h;
    em[ne].S = mb[z].front.matrix;
    em[ne].ld = mb[z].front.ld;
    em[ne].n = ns[mb[z].index];
    em[ne].rq = rq;
    em[ne].free_op = mb[z].free_op;
    emz[ne++] = z;
  }
  if (ne > 0) SC_TRY(sym_topk_group(lead, em, ne));
  for (int e = 0; e < ne; ++e) {
    // a matrix-free member with rows the candidate search could not prune: the single-call
    // path evaluates those rows exactly (its overflow words came back with the solver's syncs;
    // eig_ncluster_impl's resume branch turns the two-pass operator back on: front.free_op)
    if (em[e].free_op && em[e].status == 0 && em[e].h->h_free[0] != 0) em[e].status = 1;
  }
  const double t2 = trace ? now_us() : 0.0;
  // ---- k-means: lockstep over the solved members
  KmGroupItem km[kGroupMax];
  int kmz[kGroupMax], label_n[kGroupMax];
  size_t label_off[kGroupMax], label_total = 0;
  int nk = 0;
  {  // staging for the labels of this group (every member could end up in it)
    size_t need = 0;
    for (int z = 0; z < count; ++z) need += (size_t)ns[mb[z].index];
    // (never less than a full group of the largest utterances the grouped path takes -- 512 KB:
    //  a staging buffer that follows the batch's composition is a hipHostFree + hipHostMalloc,
    //  milliseconds, whenever a group's total grows)
    need = std::max(need, (size_t)kGroupMax * 4096);
    SC_TRY(grow(lead, lead->ginfo, (size_t)kGroupMax * 16 * sizeof(int)));
    SC_TRY(grow(lead, lead->glabels, need * sizeof(int64_t)));
    if (!lead->h_ginfo)
      SC_HIP(lead, hipHostMalloc(reinterpret_cast<void**>(&lead->h_ginfo),
                                 (size_t)kGroupMax * 16 * sizeof(int)));
    if (lead->h_glabels_count < need) {
      if (lead->h_glabels) SC_HIP(lead, hipHostFree(lead->h_glabels));
      lead->h_glabels = nullptr;
      lead->h_glabels_count = 0;
      SC_HIP(lead, hipHostMalloc(reinterpret_cast<void**>(&lead->h_glabels),
                                 need * sizeof(int64_t)));
      lead->h_glabels_count = need;
    }
  }
  for (int e = 0; e < ne; ++e) {
    Member& m = mb[emz[e]];
    if (em[e].status != 0) {
      m.state = 1;
      continue;
    }
    sc_handle h = m.h;
    const int n = em[e].n;
    int k = em[e].dc.n_clusters_raw;
    if (cfg->min_clusters > 0 && k < cfg->min_clusters) k = cfg->min_clusters;  // :295-296
    m.k = k;
    double u = 0.0;
    int trials = 0;
    std::vector<double> unused;
    if (k >= 1 && k <= 32) kmeans_seed_constants(k, &u, &trials, &unused);
    if (k < 1 || k > 32 || k > h->n_vec || n < k || cfg->max_iter <= 0 ||
        !kmeans_chain_supported(n, k, trials) || sw::kmeans_single()) {
      m.state = 1;  // the single-call path states the error or takes the other kernel
      continue;
    }
    if (diags) {
      sc_diag* dg = diags + m.index;
      dg->n = n;
      dg->n_clusters_raw = em[e].dc.n_clusters_raw;
      dg->max_delta = em[e].dc.max_delta;
      dg->eig_descending = rq.descend;
      dg->n_eigenvalues = std::min((int)em[e].w.size(), SC_MAX_EIG);
      for (int i = 0; i < dg->n_eigenvalues; ++i) dg->eigenvalues[i] = em[e].w[i];
      dg->symmetry_state = m.front.folded_rownorm ? 2 : 1;
      dg->eig_path = SC_EIG_PATH_BLOCK_LANCZOS;
      dg->eig_matvec_passes = em[e].passes;
      dg->eig_block = kEigBlock;
      dg->eig_basis = em[e].basis;
      dg->eig_max_residual = em[e].dc.max_resid;
      dg->n_clusters = k;
      if (m.free_op) {
        dg->diffuse_path = SC_DIFFUSE_PATH_FREE;
        dg->free_candidates = h->h_free[65];
        dg->free_tiles_run = h->h_free[67];
      } else if (m.front.folded_rownorm) {  // (the grouped front: the explicit product)
        dg->diffuse_path = SC_DIFFUSE_PATH_EXPLICIT;
      }
    }
    SC_TRY(ensure_kmeans(h, n));
    const int lde = round_up(n, 16);
    const double* E = ptr<double>(h->E);
    if (cfg->row_wise_renorm) {
      SC_HIP(lead, hipMemcpyAsync(h->Ek.p, h->E.p, (size_t)lde * k * sizeof(double),
                                  hipMemcpyDeviceToDevice, s));
      launch_row_renorm(s, ptr<double>(h->Ek), lde, n, k);
      E = ptr<double>(h->Ek);
    }
    KmGroupItem& it = km[nk];
    it = KmGroupItem();
    it.ET = E;
    it.lde = lde;
    it.n = n;
    it.k = k;
    it.max_iter = cfg->max_iter;
    it.first_center = sc_uniform_choice(n, u);
    it.trials = trials;
    it.ws = kmeans_workspace(h);
    it.ws.rnd = ptr<double>(lead->gkrnd) + (size_t)k * kRndStride;
    // the stop words and the labels of the whole group live side by side: one copy each
    it.ws.info = ptr<int>(lead->ginfo) + 16 * nk;
    it.ws.labels64 = ptr<long long>(lead->glabels) + label_total;
    label_off[nk] = label_total;
    label_total += (size_t)n;
    kmz[nk++] = emz[e];
  }
  int running = nk;
  for (int it = 0; running > 0; it += 4) {
    launch_kmeans_chain_group(s, km, nk, it, 4);
    SC_TRY(check_last(lead, "group kmeans launch"));
    SC_HIP(lead, hipMemcpyAsync(lead->h_ginfo, lead->ginfo.p, (size_t)nk * 16 * sizeof(int),
                                hipMemcpyDeviceToHost, s));
    SC_HIP(lead, hipStreamSynchronize(s));
    for (int q = 0; q < nk; ++q) {
      if (km[q].n <= 0 || lead->h_ginfo[16 * q + 8] == 0) continue;
      Member& m = mb[kmz[q]];
      if (diags) diags[m.index].kmeans_iterations = lead->h_ginfo[16 * q];
      m.state = 2;
      label_n[q] = km[q].n;
      km[q].n = 0;  // idle from here on
      --running;
    }
    if (running > 0 && it > cfg->max_iter + 4)
      return fail(lead, SC_ERR_HIP, "k-means chain did not reach its stop rule");
  }
  if (nk > 0) {
    SC_HIP(lead, hipMemcpyAsync(lead->h_glabels, lead->glabels.p, label_total * sizeof(int64_t),
                                hipMemcpyDeviceToHost, s));
    SC_HIP(lead, hipStreamSynchronize(s));
    for (int q = 0; q < nk; ++q)
      memcpy(labels[mb[kmz[q]].index], lead->h_glabels + label_off[q],
             (size_t)label_n[q] * sizeof(int64_t));
  }
  if (trace)
    fprintf(stderr, "[sc] group of %d (n %d..%d): eigen %.0f us, k-means %.0f us (%d members)\n",
            count, ns[mb[count - 1].index], ns[mb[0].index], t2 - t1, now_us() - t2, nk);
  // ---- members that left the common path: the single-call pipeline on their own arena
  bool any_back = false;
  for (int z = 0; z < count; ++z) any_back = any_back || mb[z].state == 1;
  // (their speculative block steps may still be running on this stream)
  if (any_back) SC_HIP(lead, hipStreamSynchronize(s));
  for (int z = 0; z < count; ++z) {
    Member& m = mb[z];
    if (m.state != 1) continue;
    // (from the refined matrix the member's front left: only the solver and k-means again)
    sc_diag local;
    sc_diag* dg = diags ? diags + m.index : &local;
    memset(dg, 0, sizeof(*dg));
    m.h->nev = 0;
    int rc = eig_ncluster_impl(m.h, cfg, dg, nullptr, &m.front);
    if (rc == SC_OK) {
      int k = dg->n_clusters_raw;
      if (cfg->min_clusters > 0 && k < cfg->min_clusters) k = cfg->min_clusters;  // :295-296
      rc = sc_cluster(m.h, cfg, k, labels[m.index], dg);
    }
    if (rc != SC_OK) {
      lead->err = m.h->err;
      return rc;
    }
  }
  return SC_OK;
}


// ---- streams on hardware queues of their own
// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by
// default; the library does not touch the process environment -- a caller may export 8 before
// its first HIP call, api.hip / INTEGRATION.md section 4; round 5's probe measured 5950-6280
// utterances/s on config 5 with the default 4 and 5750-6120 with 8: no longer a difference),
// and streams that share a queue run one after the other.  A chain's 10 us launches queued behind a front's multi-millisecond GEMM cost
// a two-lane batch 10 % (3700 instead of 4100 utterances/s on config 5) whenever the creation
// order of the process's streams (every member arena owns one, an application has its own) put
// them together.  Which queue a new stream lands on is the runtime's business (ROCm 7.2: up
// the queues, then down again -- streams created back to back do collide where it turns), so
// it is MEASURED: a one-lane kernel spins for 150 us on stream a; a marker recorded on stream
// b right after it completes at once unless b sits behind a.
__global__ void k_spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
int streams_share_queue(sc_handle h, hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb,
                        bool* share) {
  SC_HIP(h, hipStreamSynchronize(a));
  SC_HIP(h, hipStreamSynchronize(b));
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, a, 15000LL);  // 100 MHz ticks
  SC_HIP(h, hipEventRecord(ea, a));
  SC_HIP(h, hipEventRecord(eb, b));
  SC_HIP(h, hipEventSynchronize(eb));
  *share = hipEventQuery(ea) == hipSuccess;  // the spin was over before b's marker came through
  SC_HIP(h, hipStreamSynchronize(a));
  return SC_OK;
}
// Fills slots[0..count) with new streams none of which shares a hardware queue with another or
// with one of `have`; candidates that do are destroyed.  When the queues run out (a runtime
// initialised with 4 of them) the remaining slots take what comes -- the first slots, the
// chains, have been served by then.
int independent_streams(sc_handle h, std::vector<hipStream_t> have, hipStream_t** slots,
                        int count) {
  hipEvent_t ea = nullptr, eb = nullptr;
  SC_HIP(h, hipEventCreateWithFlags(&ea, hipEventDisableTiming));
  SC_HIP(h, hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  std::vector<hipStream_t> rejected;
  int filled = 0, rc = SC_OK;
  for (int tries = 0; filled < count && tries < 4 * count + 16 && rc == SC_OK; ++tries) {
    hipStream_t cand = nullptr;
    // (one priority for all of them: a lower one for the banks, so that the short kernels of
    //  the chains go first, cost 8 %, a higher one 11 % -- the GEMMs are the throughput)
    if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, 0) != hipSuccess) {
      rc = fail(h, SC_ERR_HIP, "could not create a stream for the grouped batch");
      break;
    }
    bool clash = false;
    for (size_t i = 0; i < have.size() && !clash && rc == SC_OK; ++i)
      rc = streams_share_queue(h, have[i], cand, ea, eb, &clash);
    if (rc != SC_OK || clash) {
      rejected.push_back(cand);  // (kept until the end: destroying it now would free its place)
      continue;
    }
    have.push_back(cand);
    *slots[filled++] = cand;
  }
  for (; filled < count && rc == SC_OK; ++filled)
    if (hipStreamCreateWithPriority(slots[filled], hipStreamNonBlocking, 0) != hipSuccess)
      rc = fail(h, SC_ERR_HIP, "could not create a stream for the grouped batch");
  if (sw::group_trace())
    fprintf(stderr, "[sc] streams of the batch: %d on queues of their own, %zu candidates rejected\n",
            (int)have.size(), rejected.size());
  for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
  (void)hipEventDestroy(ea);
  (void)hipEventDestroy(eb);
  return rc;
}

// One lane of the grouped batch: the groups `mine` (indices into the size-sorted list, `width`
// members each) on the lead's streams and member arenas.
int run_group_lane(sc_handle h, const double* const* xs, const int* ns, int d,
                   const sc_config* cfg, int64_t* const* labels, sc_diag* diags,
                   const std::vector<int>& grouped, const std::vector<int>& gstart, int width,
                   const std::vector<int>& mine, const EigRequest& rq, int group_limit) {
  if (mine.empty()) return SC_OK;
  // (the lead's lockstep chains run on its dedicated stream for the duration of the batch)
  struct StreamSwap {
    sc_handle h;
    hipStream_t saved;
    ~StreamSwap() { h->stream = saved; }
  } swap{h, h->stream};
  SC_HIP(h, hipStreamSynchronize(h->stream));
  h->stream = h->gchain_stream;
  SC_TRY(ensure_seed_table(h));
  memset(h->gconv_hist, 0, sizeof(h->gconv_hist));
  h->gconv_seen = 0;
  const int ngroups = (int)mine.size();
  // Banks of member arenas: while the eigensolver and k-means chains of group g run (short
  // launches, host synchronisations, Rayleigh-Ritz on the host), the GEMMs and refinement
  // passes of group g + 1 keep the chip busy on the other bank's stream.  (Two banks; a
  // third one put two more GEMMs next to the chains and cost 9 % on config 5: the chains'
  // short kernels wait longer for a free CU.)
  constexpr int banks = kGroupBanks;
  // (group g of the size-sorted list = [gstart[g], gstart[g + 1]); `width` = the largest group)
  auto group_count = [&](int j) { return gstart[mine[j] + 1] - gstart[mine[j]]; };
  auto group_members = [&](int j) { return grouped.data() + gstart[mine[j]]; };
  // arenas once.  Groups of equal cost put a member of any size at any position z (five
  // utterances of n ~ 3000 in one batch's first group, seven of n ~ 2800 in the next batch's), so
  // every position of a bank that is used at all is reserved for the largest member of the
  // lane: an arena that has to grow is a hipFree + hipMalloc in the middle of a batch (the 8-GPU
  // shares of config 5 ran 3x slower on arenas sized position by position).
  // (the largest grouped member of the BATCH, not of this lane's groups: which lane draws the
  //  large groups changes from batch to batch too)
  const int lane_largest = ns[grouped[0]];
  // (all `group_limit` positions of a bank in use, whatever this batch's largest group: the next
  //  batch -- or the next rank's share -- cuts its list elsewhere, and a position that does not
  //  exist yet is an sc_create: streams, events, pinned buffers, milliseconds)
  for (int b = 0; b < std::min(banks, ngroups); ++b)
    for (int z = 0; z < std::max(width, group_limit); ++z) {
      const int largest = lane_largest;
      sc_handle hz = nullptr;
      SC_TRY(group_slot(h, b * kGroupMax + z, &hz));  // (a stride that does not move with the batch)
      int rc = sc_reserve(hz, largest, d);
      // (... and the buffers of the matrix-free Diffuse, which a member arena otherwise grows the
      //  first time a member of n >= 1536 lands in it)
      if (rc == SC_OK && free_diffuse_wanted(hz, cfg, largest, rq, true)) rc = ensure_free(hz, largest);
      if (rc != SC_OK) return fail(h, rc, hz->err);
      hz->have_constraint = false;
    }
  Member mbs[kGroupBanks][kGroupMax];
  int front_bank[kGroupBanks];
  const bool trace = sw::group_trace();
  const bool covers = grouped_front_covers(cfg);
  if (covers) {
    SC_TRY(grow(h, h->blurw, (2 * SC_MAX_BLUR_RADIUS + 1) * sizeof(double)));
    SC_TRY(upload_blur_weights(h, cfg));
    SC_HIP(h, hipStreamSynchronize(h->stream));  // the bank streams read them
  }
  auto front = [&](int j) -> int {
    const int b = j % banks, cnt = group_count(j);
    const int* idx = group_members(j);
    // (the streaming blur of the grouped front needs every member at n >= 256; the sizes
    //  are sorted, the last member of the group is its smallest)
    if (covers && blur_group_front_supported(ns[idx[cnt - 1]], cfg->blur_radius)) {
      front_bank[b] = b;
      return enqueue_front_grouped(h, xs, ns, d, cfg, diags, idx, cnt, b * kGroupMax, mbs[b], b);
    }
    front_bank[b] = -1;
    return enqueue_front(h, xs, ns, d, cfg, diags, idx, cnt, b * kGroupMax, mbs[b]);
  };
  for (int j = 0; j < std::min(banks - 1, ngroups); ++j) SC_TRY(front(j));
  for (int j = 0; j < ngroups; ++j) {
    const double t0 = trace ? now_us() : 0.0;
    if (j + banks - 1 < ngroups) SC_TRY(front(j + banks - 1));
    if (trace) fprintf(stderr, "[sc] next front enqueued in %.0f us\n", now_us() - t0);
    SC_TRY(finish_group(h, ns, cfg, labels, diags, mbs[j % banks], group_count(j), rq,
                        front_bank[j % banks]));
  }
  return SC_OK;
}

}  // namespace

extern "C" int sc_predict_batch_grouped(sc_handle h, const double* const* xs, const int* ns,
                                        int d, int count, const sc_config* cfg,
                                        int64_t* const* labels, sc_diag* diags, int group) {
  if (!h) return SC_ERR_INVALID;
  if (!xs || !ns || !labels || count < 0) return fail(h, SC_ERR_INVALID, "NULL argument");
  SC_TRY(validate_config(h, cfg));
  SC_HIP(h, hipSetDevice(h->device));
  group = std::max(1, std::min(group, kGroupMax));
  // the batch reuses the member arenas a sweep may have left eigenvectors in
  h->sweep_slot.clear();
  for (sc_handle lane : h->glanes) lane->sweep_slot.clear();
  const EigRequest rq = make_eig_request(cfg);
  const bool cfg_ok = group > 1 && cfg->kmeans_metric == kKmeansCosine &&
                      !constraint_active(h, cfg, true) && !constraint_active(h, cfg, false);
  std::vector<int> grouped, single;
  for (int i = 0; i < count; ++i) {
    if (cfg_ok && xs[i] && ns[i] > 0 && sym_group_eligible(ns[i], rq)) grouped.push_back(i);
    else single.push_back(i);
  }
  if (!grouped.empty()) {
    // similar sizes together: a group's launches are sized by its largest member
    std::stable_sort(grouped.begin(), grouped.end(), [&](int a, int b) { return ns[a] > ns[b]; });
    // Groups: round 6 -- of equal COST, not equal count.  A group's chains advance in lockstep at
    // the pace of its slowest member and its launches are sized by its largest one; sixteen
    // utterances of n ~ 3000 are 6 ms of work, sixteen of n ~ 400 one.  With equal counts the
    // lane that drew the first group finished long after the others (the 64-utterance share of
    // an 8-GPU run: lanes at 7.4 and 4.2 ms by the cost model).  Now the size-sorted list is cut
    // where the running cost passes total / G (at most `group` members), G a multiple of the lane
    // count with at least two groups per lane (a lane overlaps the front of its next group with
    // the chains of the current one), and the groups go to the lanes longest-first, each to the
    // least loaded lane.  SC_GROUP_EQUAL_COUNT=1: round 5's slicing (A/B, profiles/r27).
    auto unit_cost = [&](int i) { const double n = ns[i]; return 60.0 + 3.5e-5 * n * n; };
    std::vector<int> gstart;
    int width = std::min(group, (int)grouped.size());
    const bool equal_count = sw::group_equal_count();
    const int min_groups = ((int)grouped.size() + width - 1) / width;
    const int lanes_wanted = std::max(
        1, std::min(kGroupLanes, grouped_front_covers(cfg) ? min_groups / kGroupBanks : 1));
    if (equal_count || min_groups < 2) {
      for (int at = 0; at < (int)grouped.size(); at += width) gstart.push_back(at);
    } else {
      double total = 0.0;
      for (int i : grouped) total += unit_cost(i);
      const int per_lane = std::max(kGroupBanks, (min_groups + lanes_wanted - 1) / lanes_wanted);
      const int target_groups = lanes_wanted * per_lane;
      const double share = total / target_groups;
      double run = 0.0;
      int start = 0;
      for (int at = 0; at < (int)grouped.size(); ++at) {
        const double c = unit_cost(grouped[at]);
        // close the group before this member if it is full, or if adding the member moves the
        // running cost further from the share than leaving it out
        if (at > start && (at - start >= width || (run + c - share > share - run))) {
          gstart.push_back(start);
          start = at;
          run = 0.0;
        }
        run += c;
      }
      gstart.push_back(start);
    }
    const int ngroups = (int)gstart.size();
    gstart.push_back((int)grouped.size());
    width = 0;
    for (int g = 0; g < ngroups; ++g) width = std::max(width, gstart[g + 1] - gstart[g]);
    // Lanes: the groups of the size-sorted list are dealt round-robin to kGroupLanes leads (the
    // caller's handle and kGroupLanes - 1 more, each with its own streams, member arenas, staging
    // and host workers), and every lead but the first is driven by its own host thread.  The
    // lockstep eigensolver / k-means chain of a group is a string of short launches, host
    // synchronisations and Rayleigh-Ritz solves on the host; with one lane the chains of all
    // groups ran end to end on one stream, and that string -- not the GEMMs -- was the batch's
    // critical path (132 ms of chain against 118 ms of GEMM / refinement kernels on config 5).
    // Measured on config 5 (utterances/s, tests/probes/group_only.py): 1 lane 3800, 2 lanes
    // 4020, 3 lanes 4200, 4 lanes 4150; the second lane's groups in ascending order of size
    // (a GEMM-heavy lane beside a latency-bound one) 3920 against 4020.
    const int lanes =
        std::max(1, std::min(kGroupLanes, grouped_front_covers(cfg) ? ngroups / kGroupBanks : 1));
    std::vector<int> lane_groups[kGroupLanes];
    if (equal_count) {
      for (int g = 0; g < ngroups; ++g) lane_groups[g % lanes].push_back(g);
    } else {  // longest group first, each to the least loaded lane (ties: the lower lane)
      std::vector<double> gcost(ngroups, 0.0);
      for (int g = 0; g < ngroups; ++g)
        for (int at = gstart[g]; at < gstart[g + 1]; ++at) gcost[g] += unit_cost(grouped[at]);
      std::vector<int> order(ngroups);
      for (int g = 0; g < ngroups; ++g) order[g] = g;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gcost[a] > gcost[b]; });
      double load[kGroupLanes] = {0.0};
      for (int g : order) {
        int best = 0;
        for (int l = 1; l < lanes; ++l)
          if (load[l] < load[best]) best = l;
        lane_groups[best].push_back(g);
        load[best] += gcost[g];
      }
    }
    sc_handle leads[kGroupLanes] = {h};
    for (int l = 1; l < lanes; ++l) {
      while ((int)h->glanes.size() < l) {
        sc_handle lane = nullptr;
        const int rc = sc_create(h->device, &lane);
        if (rc != SC_OK) return fail(h, rc, "could not create a lane of the grouped batch");
        lane->profile_level = 1;
        h->glanes.push_back(lane);
      }
      leads[l] = h->glanes[l - 1];
      leads[l]->blur_ext = h->blur_ext;
      leads[l]->err.clear();
    }
    // The streams of the batch: per lane one for the lockstep chains and one per bank for the
    // fronts, on hardware queues of their own (independent_streams above); the leads' own streams
    // idle during the batch.
    {
      hipStream_t* want[kGroupLanes * (1 + kGroupBanks)];
      int nwant = 0;
      for (int pass = 0; pass < 1 + kGroupBanks; ++pass)  // (the chains first: they matter most)
        for (int l = 0; l < lanes; ++l) {
          hipStream_t* slot =
              pass == 0 ? &leads[l]->gchain_stream : &leads[l]->gbank_stream[pass - 1];
          if (!*slot) want[nwant++] = slot;
          if (pass > 0 && !leads[l]->gbank_ev[pass - 1])
            SC_HIP(h, hipEventCreateWithFlags(&leads[l]->gbank_ev[pass - 1], hipEventDisableTiming));
        }
      if (nwant > 0) {
        std::vector<hipStream_t> have;  // (lanes of an earlier, smaller batch keep theirs)
        for (int l = 0; l < lanes; ++l) {
          if (leads[l]->gchain_stream) have.push_back(leads[l]->gchain_stream);
          for (int b2 = 0; b2 < kGroupBanks; ++b2)
            if (leads[l]->gbank_stream[b2]) have.push_back(leads[l]->gbank_stream[b2]);
        }
        SC_TRY(independent_streams(h, have, want, nwant));
      }
    }
    int rcs[kGroupLanes] = {SC_OK};
    std::vector<std::thread> side;
    bool inline_lane[kGroupLanes] = {false};
    for (int l = 1; l < lanes; ++l) {
      auto body = [&, l]() {
        if (hipSetDevice(h->device) != hipSuccess) {
          rcs[l] = fail(leads[l], SC_ERR_HIP, "hipSetDevice failed on a lane");
          return;
        }
        rcs[l] = run_group_lane(leads[l], xs, ns, d, cfg, labels, diags, grouped, gstart, width,
                                lane_groups[l], rq, equal_count ? 0 : group);
      };
      try {
        side.emplace_back(body);
      } catch (...) {  // (no thread to be had: the lane runs on this one, after lane 0)
        inline_lane[l] = true;
      }
    }
    rcs[0] = run_group_lane(h, xs, ns, d, cfg, labels, diags, grouped, gstart, width,
                            lane_groups[0], rq, equal_count ? 0 : group);
    for (int l = 1; l < lanes; ++l)
      if (inline_lane[l])
        rcs[l] = run_group_lane(leads[l], xs, ns, d, cfg, labels, diags, grouped, gstart, width,
                                lane_groups[l], rq, equal_count ? 0 : group);
    for (auto& t : side) t.join();
    {  // member arenas that hold a large share of the device do not outlive the batch (the
       // lanes keep up to 3 x 2 x 16 of them warm otherwise: 21 GB after config 5)
      size_t free_b = 0, total_b = 0, held = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        for (int l = 0; l < lanes; ++l)
          for (sc_handle sub : leads[l]->gslots) held += sub->A0.bytes + sub->B1.bytes + sub->B2.bytes;
        if (held > total_b / 4)
          for (int l = 0; l < lanes; ++l) {
            for (sc_handle sub : leads[l]->gslots) sc_destroy(sub);
            leads[l]->gslots.clear();
            leads[l]->sweep_slot.clear();
          }
      }
    }
    if (rcs[0] != SC_OK) return rcs[0];
    for (int l = 1; l < lanes; ++l)
      if (rcs[l] != SC_OK) return fail(h, rcs[l], leads[l]->err.c_str());
  }
  // (members outside the grouped path's range -- n >= 4096 above all: their uploads ride under
  //  the previous member's pipeline, api.hip predict_sequence)
  if (!single.empty())
    SC_TRY(predict_sequence(h, single.data(), (int)single.size(), xs, ns, d, cfg, labels, diags));
  return SC_OK;
}

// ------------------------------------------------------------------------------
// AutoTune: one search level as a group (reference autotune.py:98-111)
// ------------------------------------------------------------------------------
// sc_eig_ncluster for `count` values of p_percentile on the resident affinity.  The values of
// a level differ in nothing but the row threshold: CropDiagonal + GaussianBlur run once, then
// the members (one per value, up to kGroupMax per round) go through threshold + symmetrise,
// the Diffuse GEMM (all members' tiles in one launch) and the scaling vectors as grouped
// launches and through ONE lockstep block Lanczos (sym_topk_group, eigenvalues + eigengap
// decision only).  diags[i] is what sc_eig_ncluster would report for p_values[i]; no
// eigenvectors are left resident (evaluate the winner with sc_eig_ncluster).  Configurations
// the grouped stages do not cover are evaluated one by one.
extern "C" int sc_eig_ncluster_sweep(sc_handle h, const sc_config* cfg, const double* p_values,
                                     int count, sc_diag* diags) {
  if (!h) return SC_ERR_INVALID;
  if (!p_values || !diags || count < 0) return fail(h, SC_ERR_INVALID, "NULL argument");
  SC_TRY(validate_config(h, cfg));
  if (!h->have_affinity) return fail(h, SC_ERR_INVALID, "no affinity resident");
  SC_HIP(h, hipSetDevice(h->device));
  const int n = h->n, ld = h->ldn;
  const EigRequest rq = make_eig_request(cfg);
  // whatever an earlier sweep left in the member arenas is no longer adoptable (a sweep that
  // takes the one-by-one route below leaves no eigenvectors there at all)
  h->sweep_slot.clear();
  auto one_by_one = [&](int i) -> int {
    sc_config c = *cfg;
    c.p_percentile = p_values[i];
    return sc_eig_ncluster(h, &c, diags + i);
  };
  // Two sequences are swept as groups: the ICASSP2018 one, and [RowWiseThreshold, Symmetrize]
  // alone -- the Turn-to-Diarize refinement (reference configs.py:49-59: Percentile cut,
  // binarisation, preserved diagonal, Average), whose values are the thresholded affinity
  // itself: no blur to share, no Diffuse, the lockstep solver works on the symmetrised matrix.
  const bool icassp = grouped_front_covers(cfg);
  const bool thr_sym_only = cfg->n_ops == 2 && cfg->ops[0] == SC_OP_ROW_WISE_THRESHOLD &&
                            cfg->ops[1] == SC_OP_SYMMETRIZE;
  const bool grouped = count > 1 && (icassp || thr_sym_only) && h->affinity_symmetric &&
                       !constraint_active(h, cfg, false) &&
                       (!icassp || blur_group_supported(n, cfg->blur_radius)) &&
                       sym_group_eligible(n, rq, true) && !sw::sweep_one_by_one();
  if (!grouped) {
    for (int i = 0; i < count; ++i) SC_TRY(one_by_one(i));
    return SC_OK;
  }
  hipStream_t s = h->stream;
  if (icassp) {
    // ---- shared by every value: CropDiagonal's value and the blurred matrix (+ row maxima)
    const double* crop = ptr<double>(h->cropval);
    if (!h->have_cropval) {
      launch_crop_value(s, ptr<double>(h->A0), n, ld, ptr<double>(h->dvec));
      crop = ptr<double>(h->dvec);
    }
    SC_TRY(upload_blur_weights(h, cfg));
    if (!launch_gaussian_blur_fused(s, ptr<double>(h->A0), ptr<double>(h->B1), n, ld,
                                    cfg->blur_radius, ptr<double>(h->blurw), crop,
                                    ptr<double>(h->rmpart)))
      return fail(h, SC_ERR_HIP, "fused blur not taken");
    SC_TRY(check_last(h, "sweep blur launch"));
  }
  memset(h->gconv_hist, 0, sizeof(h->gconv_hist));
  h->gconv_seen = 0;
  std::vector<int> later;  // values whose solve left the common path
  h->sweep_slot.assign(count, -1);  // which member arena holds value i's eigenvectors
  h->sweep_n = n;
  h->sweep_p.assign(p_values, p_values + count);
  h->sweep_cfg = *cfg;
  // A member arena of a sweep holds the thresholded matrix (B2), its Diffuse product (B1),
  // the n-vectors and the eigensolver workspace -- no affinity copy, no k-means workspace.
  // The group is as wide as free memory allows (16 members of n = 16384 are 69 GB); whatever
  // does not fit is evaluated one value at a time on this handle, like before the grouping.
  // Matrix-free Diffuse (free_api.hip): a member then holds the thresholded matrix, its digits
  // and the fp32 tiles of their product instead of S, and the lockstep solver applies A twice.
  // The cut vector bounds max|a| when the affinity is non-negative by construction.
  const bool free_route = icassp && free_diffuse_wanted(h, cfg, n, rq);
  const bool amax_from_cut = free_route && h->affinity_from_embeddings && !h->constraint_applied &&
                             cfg->soft_multiplier >= 0.0 && cfg->soft_multiplier <= 1.0;
  const size_t member_bytes = 2 * (size_t)n * ld * sizeof(double) + (size_t)n * 8192 +
                              (free_route ? free_q_bytes(n) + free_t32_bytes(n) : 0);
  int width = 0;
  {
    size_t free_b = 0, total_b = 0;
    SC_HIP(h, hipMemGetInfo(&free_b, &total_b));
    size_t budget = (size_t)(0.85 * (double)free_b);
    for (; width < std::min(kGroupMax, count); ++width) {
      const bool ready = width < (int)h->gslots.size() &&
                         h->gslots[width]->B1.bytes >= (size_t)n * ld * sizeof(double) &&
                         h->gslots[width]->B2.bytes >= (size_t)n * ld * sizeof(double) &&
                         h->gslots[width]->Q.bytes > 0;
      if (ready) continue;
      if (budget < member_bytes) break;
      budget -= member_bytes;
    }
  }
  if (width < 2) {
    for (int i = 0; i < count; ++i) SC_TRY(one_by_one(i));
    return SC_OK;
  }
  bool out_of_memory = false;
  for (int base = 0; base < count && !out_of_memory; base += width) {
    const int cnt = std::min(width, count - base);
    FrontItem fi[kGroupMax];
    GemmGroupItem dif[kGroupMax];
    GroupEigMember em[kGroupMax];
    memset(fi, 0, sizeof(fi));
    const int nt = gemm_tile_dim(n);
    for (int z = 0; z < cnt; ++z) {
      sc_handle hz = nullptr;
      SC_TRY(group_slot(h, z, &hz));
      int rc = ensure_matrices(hz, n, 0, false);
      if (rc == SC_OK) rc = ensure_eig(hz, n);
      if (rc == SC_OK) rc = ensure_tilemap(hz, n);
      if (rc == SC_OK && free_route) rc = ensure_free(hz, n);
      if (rc == SC_ERR_OOM) {  // the estimate above was optimistic: the rest one by one
        out_of_memory = true;
        break;
      }
      if (rc != SC_OK) return fail(h, rc, hz->err);
      hz->n = n;
      hz->ldn = ld;
      hz->n_vec = 0;
      FrontItem& f = fi[z];
      f.n = n;
      f.ldn = ld;
      // the matrix every value thresholds: the shared blurred one, or the affinity itself
      f.B1 = icassp ? ptr<double>(h->B1) : ptr<double>(h->A0);  // read only
      f.B2 = ptr<double>(hz->B2);
      f.rmpart = ptr<double>(h->rmpart);
      f.blur_cols = blur_tile_columns(n, cfg->blur_radius);
      f.cut = ptr<double>(hz->cut);
      f.rowmax = ptr<double>(hz->rowmax);
      f.rowsum = ptr<double>(hz->rowsum);
      f.cvec = ptr<double>(hz->cvec);
      f.pvec = ptr<double>(hz->pvec);
      f.tvec = ptr<double>(hz->tvec);
      f.symflag = h->affinity_from_embeddings ? ptr<int>(h->symflag) : nullptr;
      f.flags = ptr<int>(hz->flags);
      f.p_own = p_values[base + z];
      dif[z] = GemmGroupItem();
      dif[z].A = f.B2;
      dif[z].lda = ld;
      dif[z].C = ptr<double>(hz->B1);
      dif[z].ldc = ld;
      dif[z].n = n;
      dif[z].K = n;
      dif[z].tilemap = hz->tilemap_cur;
      dif[z].partial_max = ptr<double>(hz->statp);
      dif[z].partial_sum = ptr<double>(hz->statp) + (size_t)n * nt;
      dif[z].rowmax = ptr<double>(hz->rowmax);
      dif[z].rowsum = ptr<double>(hz->rowsum);
      em[z] = GroupEigMember();
      em[z].h = hz;
      em[z].S = (free_route || !icassp) ? ptr<double>(hz->B2) : ptr<double>(hz->B1);
      em[z].ld = ld;
      em[z].n = n;
      em[z].rq = rq;
      em[z].free_op = free_route;
    }
    if (out_of_memory) {
      (void)hipGetLastError();
      for (int i = base; i < count; ++i) later.push_back(i);
      break;
    }
    {  // (the init kernel must not clear the shared symflag word)
      FrontItem init[kGroupMax];
      memcpy(init, fi, sizeof(init));
      for (int z = 0; z < cnt; ++z) init[z].symflag = nullptr;
      launch_front_begin_group(s, init, cnt, false);
    }
    const bool own_cuts = !icassp;  // (RowMax cuts of the ICASSP front come from the blur's partials)
    if (own_cuts) {
      if (cfg->threshold_type == SC_THRESHOLD_PERCENTILE) {
        launch_cut_percentile_group(s, fi, cnt, cfg->preserve_diagonal);  // (p_own per member)
      } else {
        for (int z = 0; z < cnt; ++z)
          launch_cut_from_rows(s, fi[z].B1, n, ld, p_values[base + z], fi[z].cut,
                               cfg->preserve_diagonal);
      }
    }
    // (round 6: with max|a| known from the cuts the threshold pass writes the members' digits
    //  itself -- free_group_prepare / free_group_digits, free_api.hip)
    FreeItem fitems[kGroupMax];
    const bool fused_digits = icassp && free_route && amax_from_cut && !sw::group_quantize_pass();
    if (fused_digits) {
      sc_handle fh0[kGroupMax];
      const double* mats[kGroupMax];
      const double* cuts[kGroupMax];
      int nn[kGroupMax], ll[kGroupMax];
      for (int z = 0; z < cnt; ++z) {
        fh0[z] = em[z].h;
        mats[z] = em[z].S;
        cuts[z] = ptr<double>(em[z].h->cut);
        nn[z] = n;
        ll[z] = ld;
      }
      if (!own_cuts) launch_cut_from_partials_group(s, fi, cnt, cfg->p_percentile);
      TsDigits digits[kGroupMax];
      SC_TRY(free_group_prepare(fh0, mats, cuts, p_values + base, cnt, ll, nn, s,
                                (cfg->binarize || cfg->preserve_diagonal) ? 1.0 : 0.0, fitems,
                                digits));
      launch_threshold_symmetrize_group(s, fi, cnt, cfg->p_percentile, cfg->soft_multiplier,
                                        cfg->binarize, cfg->symmetrize_type,
                                        cfg->preserve_diagonal, true, digits);
    } else {
      launch_threshold_symmetrize_group(s, fi, cnt, cfg->p_percentile, cfg->soft_multiplier,
                                        cfg->binarize, cfg->symmetrize_type,
                                        cfg->preserve_diagonal, own_cuts);
    }
    if (!icassp) {
      // no Diffuse: the row sums of the symmetrised matrix are the degrees
      launch_row_stats_group(s, fi, cnt);
    } else if (free_route) {
      // rowmax / rowsum of every member's S = A A^T without forming it: digits per member,
      // ONE launch for the digit products of all members, candidates + exact recheck per member
      const signed char* qs[kGroupMax];
      float* ts[kGroupMax];
      unsigned* ms[kGroupMax];
      sc_handle fh[kGroupMax];
      for (int z = 0; z < cnt; ++z) fh[z] = em[z].h;
      if (fused_digits) {
        SC_TRY(free_group_digits(fh, fitems, cnt, s));
      } else if (amax_from_cut) {
        // begin, quantiser, scan and statistics of all members: one launch each
        const double* mats[kGroupMax];
        const double* cuts[kGroupMax];
        for (int z = 0; z < cnt; ++z) {
          mats[z] = em[z].S;
          cuts[z] = ptr<double>(em[z].h->cut);
        }
        int nn[kGroupMax], ll[kGroupMax];
        for (int z = 0; z < cnt; ++z) { nn[z] = n; ll[z] = ld; }
        // (floor 1 when the pass writes ones: the same expression as the single call and the
        //  grouped batch front; grouped_front_covers() has excluded a preserved diagonal)
        SC_TRY(free_group_begin(fh, mats, cuts, p_values + base, cnt, ll, nn, s,
                                (cfg->binarize || cfg->preserve_diagonal) ? 1.0 : 0.0, fitems));
      } else {
        for (int z = 0; z < cnt; ++z) SC_TRY(free_stats_begin(em[z].h, s, em[z].S, ld, n, false));
      }
      for (int z = 0; z < cnt; ++z) {
        sc_handle hz = em[z].h;
        qs[z] = ptr<signed char>(hz->fq);
        ts[z] = ptr<float>(hz->ft32);
        ms[z] = ptr<unsigned>(hz->fwords);
      }
      {
        int nn[kGroupMax];
        const int2* tms[kGroupMax];
        const int* pls[kGroupMax];
        for (int z = 0; z < cnt; ++z) {
          nn[z] = n;
          tms[z] = em[0].h->tilemap_cur;
          pls[z] = amax_from_cut ? fitems[z].plan : nullptr;
        }
        launch_gemm_i8_sym_group(s, qs, ts, ms, cnt, nn, tms, pls);
      }
      if (amax_from_cut) {
        SC_TRY(free_group_end(fh, fitems, cnt, s));
      } else {
        for (int z = 0; z < cnt; ++z) SC_TRY(free_stats_end(em[z].h, s, em[z].S, ld, n, false));
      }
    } else {
      launch_gemm_nt_group(s, dif, cnt, kEpiNone, 1);
    }
    launch_scaling_vectors_group(s, fi, cnt, cfg->laplacian_type, icassp ? 1 : 0);
    SC_TRY(check_last(h, "sweep launch"));
    // (with the Ritz vectors: the level's winner is then adopted, not evaluated again --
    //  one upload and one launch for the whole group against a Diffuse + a solve)
    SC_TRY(sym_topk_group(h, em, cnt, true));
    if (free_route) {
      // a value with rows the candidate search could not prune (their exact evaluation needs
      // the host in the loop) goes through the single-call route, like any value that left
      // the lockstep solver: the overflow words arrived with the solver's synchronisations
      SC_HIP(h, hipStreamSynchronize(s));
      for (int z = 0; z < cnt; ++z)
        if (em[z].status == 0 && em[z].h->h_free[0] != 0) em[z].status = 1;
    }
    for (int& slot : h->sweep_slot)
      if (slot >= 0 && slot < cnt) slot = -1;  // an earlier round's member: arena reused
    for (int z = 0; z < cnt; ++z) {
      sc_diag* dg = diags + base + z;
      if (em[z].status == 0) h->sweep_slot[base + z] = z;
      if (em[z].status != 0) {
        later.push_back(base + z);  // (after the rounds: it overwrites the shared blur)
        em[z].h->eig_skip_fused = false;  // (a hint for a re-solve on that arena: none follows)
        continue;
      }
      memset(dg, 0, sizeof(*dg));
      dg->n = n;
      dg->n_clusters_raw = em[z].dc.n_clusters_raw;
      dg->max_delta = em[z].dc.max_delta;
      dg->eig_descending = rq.descend;
      dg->n_eigenvalues = std::min((int)em[z].w.size(), SC_MAX_EIG);
      for (int i = 0; i < dg->n_eigenvalues; ++i) dg->eigenvalues[i] = em[z].w[i];
      dg->symmetry_state = icassp ? 2 : 1;
      dg->eig_path = SC_EIG_PATH_BLOCK_LANCZOS;
      dg->eig_matvec_passes = em[z].passes;
      dg->eig_block = kEigBlock;
      dg->eig_basis = em[z].basis;
      dg->eig_max_residual = em[z].dc.max_resid;
      dg->diffuse_path = !icassp ? SC_DIFFUSE_PATH_NONE
                                 : (free_route ? SC_DIFFUSE_PATH_FREE : SC_DIFFUSE_PATH_EXPLICIT);
      if (free_route) {
        dg->free_candidates = em[z].h->h_free[65];
        dg->free_tiles_run = em[z].h->h_free[67];
      }
    }
  }
  SC_HIP(h, hipStreamSynchronize(s));
  {  // member arenas that hold a large share of the device do not outlive the sweep
    size_t free_b = 0, total_b = 0, held = 0;
    SC_HIP(h, hipMemGetInfo(&free_b, &total_b));
    for (sc_handle sub : h->gslots) held += sub->A0.bytes + sub->B1.bytes + sub->B2.bytes;
    if (out_of_memory || held > total_b / 4) {
      for (sc_handle sub : h->gslots) sc_destroy(sub);
      h->gslots.clear();
      h->sweep_slot.assign(count, -1);
    }
  }
  for (int i : later) SC_TRY(one_by_one(i));  // the single-call solver
  h->n_vec = 0;  // nothing of any member is resident in this handle
  h->sweep_diags.assign(diags, diags + count);
  return SC_OK;
}

// The eigenvectors of value `index` of the last sc_eig_ncluster_sweep become the handle's
// resident eigenvectors (what sc_eig_ncluster with that p_percentile would leave): a copy
// of n x cols doubles out of the member arena instead of a second refinement + Diffuse +
// eigen solve for the AutoTune winner (reference spectral_clusterer.py:274-292 keeps the
// winner's eigenvectors from the search).  `cfg`: the sweep's configuration with p_percentile =
// that value (checked).  SC_ERR_UNSUPPORTED when that value's solve left the grouped path, its
// arena has been reused or the configuration differs: evaluate it with sc_eig_ncluster then.
extern "C" int sc_sweep_adopt(sc_handle h, const sc_config* cfg, int index, sc_diag* diag) {
  if (!h || !cfg) return SC_ERR_INVALID;
  {  // the same configuration as the sweep's, at that value's p_percentile
    sc_config want = h->sweep_cfg;
    if (index >= 0 && index < (int)h->sweep_p.size()) want.p_percentile = h->sweep_p[index];
    if (h->sweep_slot.empty() || memcmp(&want, cfg, sizeof(sc_config)) != 0)
      return fail(h, SC_ERR_UNSUPPORTED, "the last sweep was not run with this configuration");
  }
  if (index < 0 || index >= (int)h->sweep_slot.size() || h->sweep_slot[index] < 0 ||
      h->sweep_slot[index] >= (int)h->gslots.size() || h->sweep_n != h->n ||
      index >= (int)h->sweep_diags.size())
    return fail(h, SC_ERR_UNSUPPORTED, "no eigenvectors of that sweep value are resident");
  SC_HIP(h, hipSetDevice(h->device));
  sc_handle hz = h->gslots[h->sweep_slot[index]];
  const int n = h->n, cols = hz->n_vec;
  if (cols < 1 || hz->n != n)
    return fail(h, SC_ERR_UNSUPPORTED, "no eigenvectors of that sweep value are resident");
  SC_TRY(ensure_eig(h, n));
  SC_HIP(h, hipMemcpyAsync(h->E.p, hz->E.p, (size_t)round_up(n, 16) * cols * sizeof(double),
                           hipMemcpyDeviceToDevice, h->stream));
  h->n_vec = cols;
  h->last_w = hz->last_w;
  if (diag) *diag = h->sweep_diags[index];
  return SC_OK;
}

