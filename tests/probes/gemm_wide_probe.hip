// Feasibility probe (not on the product path): fp64 MFMA GEMM C = A B^T with the vendor
// library's macro-tile shape -- 128 x 256 per workgroup, 4 waves x (64 x 128), ONE
// workgroup per CU (up to 512 registers per lane), triple-buffered LDS filled by direct
// global_load_lds_dwordx4 (prefetch depth 2, no VGPR staging, no ds_write).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm_wide_probe tools/gemm_wide_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 256, BK = 16, NBUF = 3;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256, 1) void k_gemm_wide(const double* __restrict__ A,
                                                     const double* __restrict__ B,
                                                     double* __restrict__ C, int n,
                                                     double* __restrict__ dbg) {
  const long long clk0 = clock64(), wall0 = wall_clock64();
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                      // [NBUF][BM * BK]
  double* Bs = smem + NBUF * BM * BK;     // [NBUF][BN * BK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_n = n / BN;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  const int row0 = ti * BM, col0 = tj * BN;

  const double* aptr[4];
  const double* bptr[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = tid + 256 * q, r = c >> 3, kc = (c & 7) ^ ((r >> 1) & 7);
    aptr[q] = A + (size_t)(row0 + r) * n + 2 * kc;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int c = tid + 256 * q, r = c >> 3, kc = (c & 7) ^ ((r >> 1) & 7);
    bptr[q] = B + (size_t)(col0 + r) * n + 2 * kc;
  }
  auto issue = [&](int buf, int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(aptr[q] + k0),
                                       (lptr_t)(As + buf * BM * BK + 2 * (256 * q + 64 * wave)),
                                       16, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(bptr[q] + k0),
                                       (lptr_t)(Bs + buf * BN * BK + 2 * (256 * q + 64 * wave)),
                                       16, 0, 0);
  };

  v4f64 acc[4][8];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 8; ++nn) acc[m][nn] = (v4f64){0.0, 0.0, 0.0, 0.0};

  const int ktiles = n / BK;
  const int arow = (wr * 64 + li) * BK;
  const int brow = (wc * 128 + li) * BK;
  const int koff0 = (2 * (0 + lg)) ^ (li & 14);
  const int koff1 = (2 * (4 + lg)) ^ (li & 14);
  double2 a0[4], b0[8], a1[4], b1[8];
#define LOAD_FRAGS(A_, B_, Ac, Bc, koff)                                                   \
  _Pragma("unroll") for (int m = 0; m < 4; ++m)                                            \
      A_[m] = *reinterpret_cast<const double2*>((Ac) + arow + m * 16 * BK + (koff));        \
  _Pragma("unroll") for (int nn = 0; nn < 8; ++nn)                                         \
      B_[nn] = *reinterpret_cast<const double2*>((Bc) + brow + nn * 16 * BK + (koff));
#define MFMA_ROWS(A_, B_, M0, M1)                                                          \
  _Pragma("unroll") for (int m = M0; m < M1; ++m) {                                        \
    _Pragma("unroll") for (int nn = 0; nn < 8; ++nn)                                       \
        acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_[m].x, B_[nn].x, acc[m][nn], 0, 0, 0); \
  }                                                                                        \
  _Pragma("unroll") for (int m = M0; m < M1; ++m) {                                        \
    _Pragma("unroll") for (int nn = 0; nn < 8; ++nn)                                       \
        acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_[m].y, B_[nn].y, acc[m][nn], 0, 0, 0); \
  }
  // software pipeline: fragments of (tile, pair) are fetched from LDS while the MFMAs of
  // the previous (tile, pair) issue; the barrier for tile kt+1 sits in the middle of the
  // MFMA stream of (kt, pair 1)
  issue(0, 0);
  if (ktiles > 1) issue(1, 1);
  if (ktiles > 2) issue(2, 2);
  if (ktiles > 2)
    __builtin_amdgcn_s_waitcnt(0x0F70 | 24);  // tile 0 landed, tiles 1 and 2 in flight
  else
    __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  LOAD_FRAGS(a0, b0, As, Bs, koff0)
  for (int kt = 0; kt < ktiles; ++kt) {
    const double* Ac = As + (kt % NBUF) * BM * BK;
    const double* Bc = Bs + (kt % NBUF) * BN * BK;
    __builtin_amdgcn_sched_barrier(0);
    LOAD_FRAGS(a1, b1, Ac, Bc, koff1)
    MFMA_ROWS(a0, b0, 0, 4)
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);  // five MFMAs
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
    MFMA_ROWS(a1, b1, 0, 2)
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < ktiles) {
      // tile kt+1 has landed once at most the loads of tile kt+2 are outstanding
      if (kt + 2 < ktiles)
        __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
      else
        __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();  // also: every wave holds its tile-kt fragments in registers
      if (kt + 3 < ktiles) issue(kt % NBUF, kt + 3);
      const double* An = As + ((kt + 1) % NBUF) * BM * BK;
      const double* Bn = Bs + ((kt + 1) % NBUF) * BN * BK;
      LOAD_FRAGS(a0, b0, An, Bn, koff0)
    }
    MFMA_ROWS(a1, b1, 2, 4)
  }
  if (dbg != nullptr && tid == 0) {
    dbg[2 * blockIdx.x] = (double)(clock64() - clk0);
    dbg[2 * blockIdx.x + 1] = (double)(wall_clock64() - wall0);
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 8; ++nn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wr * 64 + m * 16 + lg + 4 * r;
        const int col = col0 + wc * 128 + nn * 16 + li;
        C[(size_t)row * n + col] = acc[m][nn][r];
      }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192;
  std::vector<double> ha((size_t)n * n), hb((size_t)n * n);
  unsigned long long s = 88172645463325252ull;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;  // 0 random, 1 constant, 2 few-bit values
  for (size_t i = 0; i < ha.size(); ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    ha[i] = (double)(s >> 11) / 9007199254740992.0;
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    hb[i] = (double)(s >> 11) / 9007199254740992.0;
    if (mode == 1) { ha[i] = 1.0; hb[i] = 0.5; }
    if (mode == 2) { ha[i] = (double)((int)(ha[i] * 16)) / 16.0; hb[i] = (double)((int)(hb[i] * 16)) / 16.0; }
  }
  double *a, *b, *c;
  hipMalloc(&a, ha.size() * 8); hipMalloc(&b, hb.size() * 8); hipMalloc(&c, ha.size() * 8);
  hipMemcpy(a, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
  const size_t lds = sizeof(double) * NBUF * (BM + BN) * BK;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_wide),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = (n / BM) * (n / BN);
  double* dbg;
  hipMalloc(&dbg, sizeof(double) * 2 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it)
    hipLaunchKernelGGL(k_gemm_wide, dim3(blocks), dim3(256), lds, 0, a, b, c, n, dbg);
  hipDeviceSynchronize();
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  hipEventRecord(e0);
  for (int it = 0; it < reps; ++it)
    hipLaunchKernelGGL(k_gemm_wide, dim3(blocks), dim3(256), lds, 0, a, b, c, n, dbg);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  // spot check against a host dot product
  std::vector<double> hc(16);
  double maxerr = 0;
  for (int t = 0; t < 8; ++t) {
    const int i = (t * 977) % n, j = (t * 3331 + 5) % n;
    double ref = 0;
    for (int k = 0; k < n; ++k) ref += ha[(size_t)i * n + k] * hb[(size_t)j * n + k];
    double got;
    hipMemcpy(&got, c + (size_t)i * n + j, 8, hipMemcpyDeviceToHost);
    maxerr = fmax(maxerr, fabs(got - ref) / fabs(ref));
  }
  {
    std::vector<double> h(2 * blocks);
    hipMemcpy(h.data(), dbg, sizeof(double) * 2 * blocks, hipMemcpyDeviceToHost);
    double cc = 0, w = 0;
    for (int i = 0; i < blocks; ++i) { cc += h[2 * i]; w += h[2 * i + 1]; }
    printf("mode %d: shader cycles/tile %.0f (ideal %.0f), effective shader clock %.1f MHz\n", mode,
           cc / blocks, (double)n / 4 * 32 * 64, cc / w * 100.0);
  }
  printf("wide gemm n=%d: %.3f ms  %.1f TFLOP/s  (err %.1e, hip status %s)\n", n, ms,
         2.0 * n * n * n / ms / 1e9, maxerr, hipGetErrorString(hipGetLastError()));
  return 0;
}
