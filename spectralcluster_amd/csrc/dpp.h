// Cross-lane moves on the VALU (DPP) instead of the LDS crossbar (ds_bpermute, what __shfl_*
// compiles to).  The streaming blur issued 28 ds_bpermute per row and wave -- its two
// neighbour columns and a 6-step row maximum, 7.8 M wave instructions per launch at n = 8192 --
// and the digit-writing threshold pass 15 per row: the LDS pipe, not HBM, set their pace.
// gfx9 DPP controls (CDNA keeps the whole-wave shifts): wave_shr:1 / wave_shl:1 move every
// lane's value to its neighbour across all 64 lanes; row_shr:n works inside rows of 16 lanes;
// row_bcast:15 / :31 carry a row's last lane into the next row(s).  tests/probes/dpp_test.hip
// checks the semantics on the device.
#ifndef SPECTRALCLUSTER_AMD_DPP_H_
#define SPECTRALCLUSTER_AMD_DPP_H_

#include <hip/hip_runtime.h>

namespace sc {

// lanes that have no source under `CTRL` (or are masked out) receive `old`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  const long long b = __double_as_longlong(v), o = __double_as_longlong(old);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, CTRL, ROW_MASK, 0xf, false);
  const int hi =
      __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// value of lane - 1 (lane 0: its own) / lane + 1 (lane 63: its own)
__device__ __forceinline__ double lane_from_below(double v) { return dpp_f64<kDppWaveShr1>(v, v); }
__device__ __forceinline__ double lane_from_above(double v) { return dpp_f64<kDppWaveShl1>(v, v); }

// maximum over the wave, valid in lane 63 (lane 31: over lanes 0..31)
__device__ __forceinline__ double wave_max_to_last(double m) {
  m = fmax(m, dpp_f64<kDppRowShr1>(m, m));
  m = fmax(m, dpp_f64<kDppRowShr2>(m, m));
  m = fmax(m, dpp_f64<kDppRowShr4>(m, m));
  m = fmax(m, dpp_f64<kDppRowShr8>(m, m));
  m = fmax(m, dpp_f64<kDppRowBcast15, 0xa>(m, m));
  m = fmax(m, dpp_f64<kDppRowBcast31, 0xc>(m, m));
  return m;
}
// sums over each half of the wave (32 lanes), valid in lanes 31 and 63; a fixed order
__device__ __forceinline__ double half_sum_to_last(double v) {
  v += dpp_f64<kDppRowShr1>(0.0, v);
  v += dpp_f64<kDppRowShr2>(0.0, v);
  v += dpp_f64<kDppRowShr4>(0.0, v);
  v += dpp_f64<kDppRowShr8>(0.0, v);
  v += dpp_f64<kDppRowBcast15, 0xa>(0.0, v);
  return v;
}
__device__ __forceinline__ int half_sum_to_last(int v) {
  v += dpp_i32<kDppRowShr1>(0, v);
  v += dpp_i32<kDppRowShr2>(0, v);
  v += dpp_i32<kDppRowShr4>(0, v);
  v += dpp_i32<kDppRowShr8>(0, v);
  v += dpp_i32<kDppRowBcast15, 0xa>(0, v);
  return v;
}

}  // namespace sc

#endif  // SPECTRALCLUSTER_AMD_DPP_H_
