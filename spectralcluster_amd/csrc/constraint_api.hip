// Host side of the constraint operators (reference constraint.py:95-164): the resident
// constraint matrix and the GEMM chain of ConstraintPropagation; kernels in constraint.hip
// and gemm_f64.hip.
#include "handle.h"

// ------------------------------------------------------------------------------
// N3: constraints (reference constraint.py:95-164)
// ------------------------------------------------------------------------------
// exact symmetry of a resident (n, ld) matrix; one 4-byte D2H + stream sync
int device_is_symmetric(sc_handle h, const double* m, int n, int ld, bool* out) {
  SC_TRY(grow(h, h->symflag, 16));
  const int one = 1;
  int result = 0;
  SC_HIP(h, hipMemcpyAsync(h->symflag.p, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
  launch_symmetry_flag(h->stream, m, n, ld, ptr<int>(h->symflag));
  SC_HIP(h, hipMemcpyAsync(&result, h->symflag.p, sizeof(int), hipMemcpyDeviceToHost,
                           h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  *out = result != 0;
  return SC_OK;
}

// ConstraintPropagation.adjust_affinity (constraint.py:138-164):  out may alias a.
//   P = alpha D^-1/2 A D^-1/2,  T = (I - P)^-1 = prod_{j>=0} (I + P^(2^j))  (rho(P) <= |alpha|),
//   F = (1 - alpha)^2 T Q T,  out = F > 0 ? 1 - (1 - F)(1 - A) : (1 + F) A.
// Every product runs on the fp64 MFMA GEMM (C = X Y^T).  For a symmetric A all factors
// are symmetric and commute, so squarings and T updates compute the upper tile triangle
// only; a general A carries explicit transposes instead.
static int constraint_propagation(sc_handle h, const double* a, bool sym_a, const double* q,
                                  bool sym_q, double alpha, double* out, int n, int ld) {
  hipStream_t s = h->stream;
  const double mag = fabs(alpha);
  if (!(mag < 1.0))
    return fail(h, SC_ERR_UNSUPPORTED,
                "ConstraintPropagation on the device needs |constraint_propagation_alpha| < 1");
  // factors (I + P^(2^j)), j = 0 .. steps-1, leave a remainder of P^(2^steps)
  int steps = 0;
  if (mag > 0.0) {
    double rem = mag;
    while (rem > 1e-18 && steps < 18) {
      rem *= rem;
      ++steps;
    }
    if (rem > 1e-18)
      return fail(h, SC_ERR_UNSUPPORTED,
                  "constraint_propagation_alpha too close to 1 for the Neumann product");
  }
  const size_t bytes = (size_t)n * ld * sizeof(double);
  for (int i = 0; i < 5; ++i) SC_TRY(grow(h, h->cp[i], bytes));
  SC_TRY(ensure_tilemap(h, n));
  double* P = ptr<double>(h->cp[0]);
  double* T = ptr<double>(h->cp[1]);
  double* Pn = ptr<double>(h->cp[2]);
  double* Tn = ptr<double>(h->cp[3]);
  double* X = ptr<double>(h->cp[4]);  // transposes (general A), then T Q^T
  double* ws = ptr<double>(h->splitk);
  const int2* tm = h->tilemap_cur;
  launch_row_stats(s, a, n, ld, ptr<double>(h->cut), ptr<double>(h->deg));  // deg = rowsum
  launch_cp_prepare(s, a, ptr<double>(h->deg), alpha, P, T, n, ld);          // T = I + P
  for (int j = 1; j < steps; ++j) {
    // Pn = P P
    if (sym_a) {
      launch_gemm_nt(s, P, ld, P, ld, Pn, ld, n, n, n, kEpiNone, true, ws, tm);
    } else {
      launch_transpose(s, P, X, n, ld);
      launch_gemm_nt(s, P, ld, X, ld, Pn, ld, n, n, n, kEpiNone, false, ws, nullptr);
    }
    std::swap(P, Pn);
    // Tn = T + T P
    if (sym_a) {
      launch_gemm_nt(s, T, ld, P, ld, Tn, ld, n, n, n, kEpiAdd, true, ws, tm, nullptr, T);
    } else {
      launch_transpose(s, P, X, n, ld);
      launch_gemm_nt(s, T, ld, X, ld, Tn, ld, n, n, n, kEpiAdd, false, ws, nullptr, nullptr, T);
    }
    std::swap(T, Tn);
  }
  // G^T = T^T Q^T  (X),  T Q T = T (G^T)^T  (Pn)
  const double* Tt = T;
  if (!sym_a) {
    launch_transpose(s, T, Tn, n, ld);
    Tt = Tn;
  }
  launch_gemm_nt(s, Tt, ld, q, ld, X, ld, n, n, n, kEpiNone, false, ws, nullptr);
  const bool sym_f = sym_a && sym_q;
  launch_gemm_nt(s, T, ld, X, ld, Pn, ld, n, n, n, kEpiNone, sym_f, ws, sym_f ? tm : nullptr);
  launch_cp_adjust(s, Pn, a, (1.0 - alpha) * (1.0 - alpha), out, n, ld);
  return check_last(h, "constraint propagation launch");
}

// cfg's constraint operator on `a` with the resident constraint matrix; out may alias a
int adjust_affinity(sc_handle h, const sc_config* cfg, const double* a, bool sym_a,
                           double* out, int n, int ld) {
  if (cfg->constraint_name == SC_CONSTRAINT_AFFINITY_INTEGRATION) {
    if (cfg->integration_type != SC_INTEGRATION_MAX &&
        cfg->integration_type != SC_INTEGRATION_AVERAGE)
      return fail(h, SC_ERR_INVALID, "Unsupported integration type");
    launch_affinity_integration(h->stream, a, ptr<double>(h->Cq), out, n, ld,
                                cfg->integration_type);
    return check_last(h, "affinity integration launch");
  }
  if (cfg->constraint_name == SC_CONSTRAINT_PROPAGATION)
    return constraint_propagation(h, a, sym_a, ptr<double>(h->Cq), h->constraint_symmetric,
                                  cfg->constraint_alpha, out, n, ld);
  return fail(h, SC_ERR_INVALID, "constraint_name must be a ConstraintName");
}

extern "C" int sc_set_constraint(sc_handle h, const double* q, int n) {
  if (!h) return SC_ERR_INVALID;
  if (!q || n <= 0) return fail(h, SC_ERR_INVALID, "constraint matrix must be (n, n)");
  SC_HIP(h, hipSetDevice(h->device));
  const int ld = matrix_ld(n);
  SC_TRY(grow(h, h->Cq, (size_t)n * ld * sizeof(double)));
  SC_TRY(h2d_matrix(h, q, n, n, ptr<double>(h->Cq), ld));
  SC_TRY(device_is_symmetric(h, ptr<double>(h->Cq), n, ld, &h->constraint_symmetric));
  h->have_constraint = true;
  h->qn = n;
  return SC_OK;
}

extern "C" int sc_clear_constraint(sc_handle h) {
  if (!h) return SC_ERR_INVALID;
  h->have_constraint = false;
  h->qn = 0;
  return SC_OK;
}

bool constraint_active(sc_handle h, const sc_config* cfg, bool before) {
  return cfg->constraint_name != SC_CONSTRAINT_NONE && h->have_constraint &&
         (cfg->constraint_before_refinement != 0) == before;
}

extern "C" int sc_apply_constraint(sc_handle h, const sc_config* cfg) {
  if (!h) return SC_ERR_INVALID;
  SC_TRY(validate_config(h, cfg));
  if (!h->have_affinity) return fail(h, SC_ERR_INVALID, "no affinity resident");
  if (!h->have_constraint) return fail(h, SC_ERR_INVALID, "no constraint matrix resident");
  h->sweep_slot.clear();  // (a sweep on the unadjusted affinity)
  if (cfg->constraint_name == SC_CONSTRAINT_NONE)
    return fail(h, SC_ERR_INVALID, "no constraint operation configured");
  if (h->qn != h->n)
    return fail(h, SC_ERR_INVALID, "affinity and constraint matrix must have the same shape");
  if (h->constraint_applied)
    return fail(h, SC_ERR_INVALID, "the resident affinity is already constraint-adjusted");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, h->n, 0));
  SC_TRY(adjust_affinity(h, cfg, ptr<double>(h->A0), h->affinity_symmetric, ptr<double>(h->A0),
                         h->n, h->ldn));
  h->affinity_symmetric = h->affinity_symmetric && h->constraint_symmetric;
  h->have_cropval = false;
  h->constraint_applied = true;
  h->n_vec = 0;
  return SC_OK;
}

extern "C" int sc_stage_constraint(sc_handle h, const sc_config* cfg, const double* affinity,
                                   const double* q, int n, double* out) {
  if (!h) return SC_ERR_INVALID;
  SC_TRY(validate_config(h, cfg));
  if (!affinity || !q || !out || n <= 0)
    return fail(h, SC_ERR_INVALID, "affinity and constraint matrix must be (n, n)");
  SC_TRY(sc_set_affinity(h, affinity, n));
  SC_TRY(sc_set_constraint(h, q, n));
  const int rc = sc_apply_constraint(h, cfg);
  sc_clear_constraint(h);
  SC_TRY(rc);
  return d2h_matrix(h, ptr<double>(h->A0), h->ldn, n, n, out);
}

