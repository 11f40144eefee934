// R2: GaussianBlur (reference refinement.py:160-162 -> scipy.ndimage.
// gaussian_filter, scipy 1.15.3): separable correlation, axis 0 then axis 1,
// boundary `reflect` (d c b a | a b c d | d c b a), symmetric-kernel summation
// order of scipy's NI_Correlate1D:
//     t = x[0] * w[0];  for j = radius .. 1:  t += (x[-j] + x[+j]) * w[j]
// Both passes are fused in one kernel: a (TH+2R) x (TW+2R) halo tile is staged
// in LDS, the axis-0 pass writes a TH x (TW+2R) intermediate (full fp64, like
// scipy's float64 intermediate array), the axis-1 pass writes the output tile.
// HBM traffic: 1 read (+ halo re-reads served by L2) + 1 write of n^2.
// Compiled with -ffp-contract=off so mul/add round separately like the C code.
#include <algorithm>
#include <cstring>

#include "dpp.h"
#include "sc_internal.h"

namespace sc {

constexpr int TH = 32;
constexpr int TW = 64;

__device__ __forceinline__ int reflect_index(int j, int n) {
  // scipy "reflect": period 2n, second half mirrored, edge sample repeated
  const int period = 2 * n;
  j %= period;
  if (j < 0) j += period;
  return j < n ? j : period - 1 - j;
}

__global__ __launch_bounds__(256) void k_gaussian_blur(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    int radius, const double* __restrict__ weights) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int R = radius;
  const int HW = TW + 2 * R;          // halo tile width
  const int HH = TH + 2 * R;          // halo tile height
  double* tile = smem;                 // HH x HW
  double* mid = smem + HH * HW;        // TH x HW
  double* w = mid + TH * HW;           // R + 1 : w[j] = weight at distance j

  const int i0 = blockIdx.y * TH;
  const int j0 = blockIdx.x * TW;
  const int tid = threadIdx.x;

  if (tid <= R) w[tid] = weights[R - tid];  // symmetric: left half reversed
  for (int e = tid; e < HH * HW; e += 256) {
    const int r = e / HW, c = e - r * HW;
    const int gi = reflect_index(i0 + r - R, n);
    const int gj = reflect_index(j0 + c - R, n);
    tile[e] = in[(size_t)gi * ld + gj];
  }
  __syncthreads();
  // axis 0 (down the rows)
  for (int e = tid; e < TH * HW; e += 256) {
    const int r = e / HW, c = e - r * HW;
    const double* x = tile + (r + R) * HW + c;
    double t = x[0] * w[0];
    for (int j = R; j >= 1; --j) t += (x[-j * HW] + x[j * HW]) * w[j];
    mid[e] = t;
  }
  __syncthreads();
  // axis 1 (along the row)
  for (int e = tid; e < TH * TW; e += 256) {
    const int r = e / TW, c = e - r * TW;
    const int gi = i0 + r, gj = j0 + c;
    if (gi < n && gj < n) {
      const double* x = mid + r * HW + c + R;
      double t = x[0] * w[0];
      for (int j = R; j >= 1; --j) t += (x[-j] + x[j]) * w[j];
      out[(size_t)gi * ld + gj] = t;
    }
  }
}

// Fast path (radius R known at compile time, n >= 128).  One workgroup = 4 waves
// produces a 64-row x (64 - 2R)-column output tile:
//   * the (64 + 2R) x 64 halo tile is loaded once, one 512-byte row per wave
//     instruction, into LDS (the only LDS traffic);
//   * axis 0: lane = column, each wave owns 16 output rows and reads its 16 + 2R halo
//     rows from LDS ONCE into registers (2.5 LDS reads per output instead of 2R + 1);
//   * axis 1: the 64 halo columns of a row sit in the 64 lanes of the wave, so the
//     horizontal taps are wave shuffles -- no second LDS buffer;
//   * lanes R .. 63-R hold valid outputs: one coalesced store per row, and the row
//     maximum (for RowWiseThreshold) is a wave reduction.
// scipy's summation order is kept in both passes:  t = x0 w0; t += (x[-j] + x[+j]) w[j],
// j = R .. 1.  Optional fusions:
//   diag   != nullptr : element (i, i) of the input is replaced by diag[i] on load
//                       (CropDiagonal folded in: reference refinement.py:148-150)
//   rowmax != nullptr : per-tile row maxima of the OUTPUT are written to
//                       rowmax[row * gridDim.x + blockIdx.x]
constexpr int kBlurRows = 64;
template <int R>
__global__ __launch_bounds__(256) void k_gaussian_blur_r(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    const double* __restrict__ weights, const double* __restrict__ diag,
    double* __restrict__ rowmax) {
  constexpr int HH = kBlurRows + 2 * R;
  constexpr int OW = 64 - 2 * R;      // output columns per tile
  constexpr int WR = kBlurRows / 4;   // output rows per wave
  __shared__ double tile[HH * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i0 = blockIdx.y * kBlurRows;
  const int j0 = blockIdx.x * OW;
  double w[R + 1];
#pragma unroll
  for (int j = 0; j <= R; ++j) w[j] = weights[R - j];  // symmetric: w[j] = weight at +-j

  // reflected global column of this lane (scipy "reflect": edge sample repeated)
  int gj = j0 - R + lane;
  gj = gj < 0 ? -gj - 1 : gj;
  gj = gj >= n ? 2 * n - 1 - gj : gj;
  gj = gj < 0 ? 0 : gj;  // overhang lanes of the last tile column (never stored)
  for (int r = wv; r < HH; r += 4) {
    int gi = i0 - R + r;
    gi = gi < 0 ? -gi - 1 : gi;
    gi = gi >= n ? 2 * n - 1 - gi : gi;
    gi = gi < 0 ? 0 : gi;
    double v = in[(size_t)gi * ld + gj];
    if (diag != nullptr && gi == gj) v = diag[gi];
    tile[r * 64 + lane] = v;
  }
  __syncthreads();

  // axis 0: rows wv*WR .. wv*WR + WR - 1 of the tile's output, column = lane
  double x[WR + 2 * R];
#pragma unroll
  for (int q = 0; q < WR + 2 * R; ++q) x[q] = tile[(wv * WR + q) * 64 + lane];
  const int gjo = j0 + lane - R;                 // global column of this lane's output
  const bool col_ok = lane >= R && lane < 64 - R && gjo < n;
#pragma unroll
  for (int o = 0; o < WR; ++o) {
    double t = x[o + R] * w[0];
#pragma unroll
    for (int j = R; j >= 1; --j) t += (x[o + R - j] + x[o + R + j]) * w[j];
    // axis 1 on this row: neighbours live in the neighbouring lanes
    double u = t * w[0];
#pragma unroll
    for (int j = R; j >= 1; --j) u += (__shfl_up(t, j) + __shfl_down(t, j)) * w[j];
    const int gi = i0 + wv * WR + o;
    const bool ok = col_ok && gi < n;
    if (ok) out[(size_t)gi * ld + gjo] = u;
    if (rowmax != nullptr) {
      double m = ok ? u : -INFINITY;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
      if (lane == 0 && gi < n) rowmax[(size_t)gi * gridDim.x + blockIdx.x] = m;
    }
  }
}

// Streaming path for large matrices (n >= kStreamMinN): no LDS, no barrier.  A wave walks
// down a strip of 256 input columns (4 adjacent columns per lane: one 32-byte load per lane
// and row, 2 KiB contiguous per wave) over 64 output rows, keeping the 2R+1 rows of the
// vertical stencil -- plus kAhead rows already in flight -- in a register ring whose slot
// indices are compile-time (the row loop is unrolled by the ring size).  The horizontal taps
// of a lane's 4 outputs come from its own registers and its R/4 neighbour lanes on each side
// (2R/4 shuffles of 4 doubles per 4 outputs, against 2R per output in the tile kernel above);
// stores are 32 contiguous bytes per lane.  Read amplification: (64 + 2R)/64 vertically,
// 256/(256 - 2R) horizontally.  Same summation order as scipy in both passes.
//
// Rows per wave are chosen per launch so that the grid fills the chip in WHOLE rounds: the
// kernel keeps 165 (R = 4) / 252 (R = 8) VGPRs, i.e. 3 / 2 workgroups per CU = 768 / 512
// resident workgroups.  The fixed 64 rows of round 2 made 1088 workgroups at n = 8192 -- a
// full round and a 42 % one behind it, each wave a dependent chain of row loads (318 us,
// 0.42 of the HBM peak, VALUBusy 29 %) -- and 272 at n = 4096, a third of the slots.  Same
// arithmetic per output row whatever the split.
constexpr int kStreamMinN = 512;
constexpr int kStreamRowsMin = 16;   // fewer rows per wave: the 2 R halo rows dominate
constexpr int kStreamRowsMax = 128;
constexpr int kAhead = 3;         // rows loaded ahead of the vertical stencil
template <int R>
__device__ __forceinline__ void gaussian_blur_stream_body(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    const double* __restrict__ weights, const double* __restrict__ diag,
    double* __restrict__ rowmax, const int bx, const int by, const int ncols,
    const int rows_per_wave) {
  static_assert(R % 4 == 0, "neighbour lanes carry 4 columns each");
  constexpr int S = 2 * R + 1 + kAhead;  // ring slots
  constexpr int NB = R / 4;              // neighbour lanes per side
  constexpr int OW = 256 - 2 * R;        // output columns per strip
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = bx * OW;                                // first output column
  const int r0 = (by * 4 + wv) * rows_per_wave;          // first output row
  if (r0 >= n) return;
  const int rend = min(n, r0 + rows_per_wave);
  double w[R + 1];
#pragma unroll
  for (int j = 0; j <= R; ++j) w[j] = weights[R - j];
  // the lane's 4 input columns (scipy "reflect"), and whether the whole strip is interior
  const int q0 = 4 * lane;
  int gjin[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int g = j0 - R + q0 + e;
    g = g < 0 ? -g - 1 : g;
    g = g >= n ? 2 * n - 1 - g : g;
    gjin[e] = g < 0 ? 0 : g;  // overhang of the last strip (never stored)
  }
  const bool interior = j0 - R >= 0 && j0 - R + 255 < n;
  auto load_row = [&](int r, double (&dst)[4]) {
    int gi = r;
    gi = gi < 0 ? -gi - 1 : gi;
    gi = gi >= n ? 2 * n - 1 - gi : gi;
    gi = gi < 0 ? 0 : (gi >= n ? n - 1 : gi);
    const double* row = in + (size_t)gi * ld;
    if (interior) {
      const double2 lo = *reinterpret_cast<const double2*>(row + j0 - R + q0);
      const double2 hi = *reinterpret_cast<const double2*>(row + j0 - R + q0 + 2);
      dst[0] = lo.x; dst[1] = lo.y; dst[2] = hi.x; dst[3] = hi.y;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = row[gjin[e]];
    }
    if (diag != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gjin[e] == gi) dst[e] = diag[gi];
    }
  };
  double win[S][4];
#pragma unroll
  for (int q = 0; q < S - 1; ++q) load_row(r0 - R + q, win[q]);
  // output column of (lane, e): j0 + q0 + e - R, valid for R <= q0 + e < 256 - R
  const int gjo0 = j0 + q0 - R;
  const bool lane_ok = q0 >= R && q0 + 3 < 256 - R;
  const bool full_store = lane_ok && gjo0 + 3 < n;
  for (int o = r0; o < rend; o += S) {
#pragma unroll
    for (int u = 0; u < S; ++u) {
      const int gi = o + u;
      if (gi >= rend) break;  // wave-uniform
      load_row(gi + R + kAhead, win[(u + S - 1) % S]);
      double t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double acc = win[(u + R) % S][e] * w[0];
#pragma unroll
        for (int j = R; j >= 1; --j)
          acc += (win[(u + R - j) % S][e] + win[(u + R + j) % S][e]) * w[j];
        t[e] = acc;
      }
      // horizontal: own 4 values + NB neighbour lanes on each side
      double ext[(2 * NB + 1) * 4];
#pragma unroll
      for (int e = 0; e < 4; ++e) ext[NB * 4 + e] = t[e];
      // (neighbour lanes by whole-wave DPP shifts, not ds_bpermute: dpp.h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double below = t[e], above = t[e];
#pragma unroll
        for (int d = 1; d <= NB; ++d) {
          below = lane_from_below(below);
          above = lane_from_above(above);
          ext[(NB - d) * 4 + e] = below;
          ext[(NB + d) * 4 + e] = above;
        }
      }
      double res[4];
      double m = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double acc = ext[NB * 4 + e] * w[0];
#pragma unroll
        for (int j = R; j >= 1; --j)
          acc += (ext[NB * 4 + e - j] + ext[NB * 4 + e + j]) * w[j];
        res[e] = acc;
        if (lane_ok && gjo0 + e < n) m = fmax(m, acc);
      }
      double* orow = out + (size_t)gi * ld + gjo0;
      if (out == nullptr) {
        // (row maxima only: tests/probes/blur_probe.hip measures what the write costs)
      } else if (full_store) {
        *reinterpret_cast<double2*>(orow) = make_double2(res[0], res[1]);
        *reinterpret_cast<double2*>(orow + 2) = make_double2(res[2], res[3]);
      } else if (lane_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gjo0 + e < n) orow[e] = res[e];
      }
      if (rowmax != nullptr) {
        m = wave_max_to_last(m);
        if (lane == 63) rowmax[(size_t)gi * ncols + bx] = m;
      }
    }
  }
}
template <int R>
__global__ __launch_bounds__(256) void k_gaussian_blur_stream(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld,
    const double* __restrict__ weights, const double* __restrict__ diag,
    double* __restrict__ rowmax, int rows) {
  gaussian_blur_stream_body<R>(in, out, n, ld, weights, diag, rowmax, blockIdx.x, blockIdx.y,
                               gridDim.x, rows);
}
// grouped form (batch_group.hip): blockIdx.z = member of a batch group; A0 -> B1 with the
// CropDiagonal value applied on load, per-strip row maxima into rmpart
template <int R>
__global__ __launch_bounds__(256) void k_gaussian_blur_stream_g(const GroupOf<FrontItem> g,
                                                                const double* __restrict__ weights,
                                                                int rows) {
  const FrontItem& a = g.s[blockIdx.z];
  if ((int)blockIdx.x >= a.blur_cols || (int)blockIdx.y * 4 * rows >= a.n) return;
  gaussian_blur_stream_body<R>(a.A0, a.B1, a.n, a.ldn, weights, a.cropval, a.rmpart,
                               blockIdx.x, blockIdx.y, a.blur_cols, rows);
}

// Rows per wave for `wave_rows_total` = sum over the launch's matrices of strips * n (one
// wave-row = one output row of one 256-column strip): the smallest number of whole rounds
// of resident workgroups whose share per wave stays within [kStreamRowsMin, kStreamRowsMax].
static int stream_rows_per_wave(long long wave_rows_total, int radius) {
  const long long slots = 256LL * (radius == 4 ? 3 : 2) * 4;  // resident waves on the chip
  for (int rounds = 1;; ++rounds) {
    const long long rows = (wave_rows_total + slots * rounds - 1) / (slots * rounds);
    if (rows <= kStreamRowsMax) return (int)std::max<long long>(rows, kStreamRowsMin);
  }
}

__global__ void k_copy_matrix(const double* __restrict__ in,
                              double* __restrict__ out, int n, int ld) {
  const size_t total = (size_t)n * ld;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x)
    out[e] = in[e];
}

// Any radius (sigma > 8: radius > SC_MAX_BLUR_RADIUS, up to many times n): the separable
// correlation as scipy runs it -- the whole matrix along axis 0, then along axis 1
// (scipy/ndimage/_filters.py gaussian_filter; refinement.py:160-162) -- one thread per output
// element, same summation order (centre tap, then the pairs from the outermost inwards),
// "reflect" extension with as many reflections as the radius needs.  O(n^2 r) loads from
// the caches; a path for completeness, not for speed (sigma = 1 in every preset).
template <int AXIS>
__global__ __launch_bounds__(256) void k_gaussian_blur_axis(
    const double* __restrict__ in, double* __restrict__ out, int n, int ld, int radius,
    const double* __restrict__ weights) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n) return;
  double t = in[(size_t)i * ld + j] * weights[radius];
  for (int q = radius; q >= 1; --q) {
    double lo, hi;
    if (AXIS == 0) {
      lo = in[(size_t)reflect_index(i - q, n) * ld + j];
      hi = in[(size_t)reflect_index(i + q, n) * ld + j];
    } else {
      lo = in[(size_t)i * ld + reflect_index(j - q, n)];
      hi = in[(size_t)i * ld + reflect_index(j + q, n)];
    }
    t += (lo + hi) * weights[radius - q];
  }
  out[(size_t)i * ld + j] = t;
}

void launch_gaussian_blur_any_radius(hipStream_t s, const double* in, double* tmp, double* out,
                                     int n, int ld, int radius, const double* weights_dev) {
  dim3 grid((n + 255) / 256, n);
  hipLaunchKernelGGL(k_gaussian_blur_axis<0>, grid, dim3(256), 0, s, in, tmp, n, ld, radius,
                     weights_dev);
  hipLaunchKernelGGL(k_gaussian_blur_axis<1>, grid, dim3(256), 0, s, tmp, out, n, ld, radius,
                     weights_dev);
}

void launch_gaussian_blur(hipStream_t s, const double* in, double* out, int n,
                          int ld, int radius, const double* weights_dev) {
  launch_gaussian_blur_fused(s, in, out, n, ld, radius, weights_dev, nullptr, nullptr);
}

// tile columns of the fast path for `radius` (row-max partials per row)
int blur_tile_columns(int n, int radius) {
  const int ow = (n >= kStreamMinN ? 256 : 64) - 2 * radius;
  return (n + ow - 1) / ow;
}

// Returns true if the row-max partials were produced (fast path taken).
bool launch_gaussian_blur_fused(hipStream_t s, const double* in, double* out, int n, int ld,
                                int radius, const double* weights_dev, const double* diag,
                                double* rowmax_partials) {
  dim3 grid((n + TW - 1) / TW, (n + TH - 1) / TH);
  if ((radius == 4 || radius == 8) && n >= kStreamMinN) {
    // A wave walks its rows one after the other (a chain of dependent row loads); the rows
    // per wave make the grid a whole number of rounds of resident workgroups (above)
    const int strips = blur_tile_columns(n, radius);
    int rows = stream_rows_per_wave((long long)strips * n, radius);
    // (the grid is strips x ceil(n / (4 rows)) workgroups: the rounding of the second factor
    //  must not push it past the round)
    const int per_round = 256 * (radius == 4 ? 3 : 2);
    while (rows < kStreamRowsMax) {
      const long long wgs = (long long)strips * ((n + 4 * rows - 1) / (4 * rows));
      if (wgs <= per_round || wgs % per_round == 0 || wgs % per_round > per_round * 3 / 4) break;
      ++rows;
    }
    dim3 sgrid(strips, (n + 4 * rows - 1) / (4 * rows));
    if (radius == 4)
      hipLaunchKernelGGL((k_gaussian_blur_stream<4>), sgrid, dim3(256), 0, s, in, out, n, ld,
                         weights_dev, diag, rowmax_partials, rows);
    else
      hipLaunchKernelGGL((k_gaussian_blur_stream<8>), sgrid, dim3(256), 0, s, in, out, n, ld,
                         weights_dev, diag, rowmax_partials, rows);
    return rowmax_partials != nullptr;
  }
  if ((radius == 4 || radius == 8) && n >= 128) {
    dim3 fgrid(blur_tile_columns(n, radius), (n + kBlurRows - 1) / kBlurRows);
    if (radius == 4)
      hipLaunchKernelGGL((k_gaussian_blur_r<4>), fgrid, dim3(256), 0, s, in, out, n, ld,
                         weights_dev, diag, rowmax_partials);
    else
      hipLaunchKernelGGL((k_gaussian_blur_r<8>), fgrid, dim3(256), 0, s, in, out, n, ld,
                         weights_dev, diag, rowmax_partials);
    return rowmax_partials != nullptr;
  }
  // generic path: no fusion (the caller materialises CropDiagonal itself)
  if (radius <= 0) {  // sigma == 0: gaussian_filter degenerates to a copy
    hipLaunchKernelGGL(k_copy_matrix, dim3(2048), dim3(256), 0, s, in, out, n, ld);
    return false;
  }
  const int HW = TW + 2 * radius, HH = TH + 2 * radius;
  const size_t lds = sizeof(double) * ((size_t)HH * HW + (size_t)TH * HW + radius + 1);
  SC_OPT_IN_LDS(k_gaussian_blur, 150 * 1024);
  hipLaunchKernelGGL(k_gaussian_blur, grid, dim3(256), lds, s, in, out, n, ld, radius,
                     weights_dev);
  return false;
}

// The streaming blur of every member of a batch group in one launch.  (kStreamMinN is where a
// SINGLE matrix has enough column strips for the streaming kernel to beat the tile kernel; an
// AutoTune sweep blurs its one matrix with the single-call launcher, hence the same bound.)
bool blur_group_supported(int n_min, int radius) {
  return (radius == 4 || radius == 8) && n_min >= kStreamMinN;
}
// Grouped FRONT of a batch: the strips of 16 members fill the chip whatever their size, so the
// streaming kernel takes every member of at least one full strip height -- config 5's 41
// utterances below n = 512 no longer run their stages member by member.
constexpr int kStreamGroupMinN = 256;
bool blur_group_front_supported(int n_min, int radius) {
  return (radius == 4 || radius == 8) && n_min >= kStreamGroupMinN;
}
// strips per row of the streaming kernel (row-max partials per row), whatever n
int blur_stream_columns(int n, int radius) {
  const int ow = 256 - 2 * radius;
  return (n + ow - 1) / ow;
}
void launch_gaussian_blur_group(hipStream_t s, const FrontItem* items, int count, int radius,
                                const double* weights_dev) {
  GroupOf<FrontItem> g;
  memset(&g, 0, sizeof(g));
  int nmax = 0, cmax = 0;
  for (int z = 0; z < count; ++z) {
    g.s[z] = items[z];
    nmax = std::max(nmax, items[z].n);
    cmax = std::max(cmax, items[z].n > 0 ? items[z].blur_cols : 0);
  }
  if (nmax == 0) return;
  // rows per wave from the work of the whole group (whole rounds of resident workgroups)
  long long total = 0;
  for (int z = 0; z < count; ++z)
    if (items[z].n > 0) total += (long long)items[z].blur_cols * items[z].n;
  const int rows = stream_rows_per_wave(total, radius);
  dim3 grid(cmax, (nmax + 4 * rows - 1) / (4 * rows), count);
  if (radius == 4)
    hipLaunchKernelGGL((k_gaussian_blur_stream_g<4>), grid, dim3(256), 0, s, g, weights_dev,
                       rows);
  else
    hipLaunchKernelGGL((k_gaussian_blur_stream_g<8>), grid, dim3(256), 0, s, g, weights_dev,
                       rows);
}

}  // namespace sc
