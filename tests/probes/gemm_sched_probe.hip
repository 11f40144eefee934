// Scheduling probe (not on the product path): the main loop of csrc/gemm_f64.hip (128 x 128
// tile, 4 waves, two workgroups per CU, paired-k ds_read_b128 fragments) on a full
// n x n x n product, with different wave-priority schemes between the two co-resident
// workgroups of a CU.  Question: is the ~16 % gap to the MFMA peak the two workgroups
// running in phase (both in their barrier / LDS-refill section at the same time)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm_sched_probe tools/gemm_sched_probe.hip
//   tools/bin/gemm_sched_probe [n] [reps] [ld]
// SCHED 0: product scheme (priority 1 in the MFMA section, 0 elsewhere)
//       1: "late": first half of the MFMA section priority 1, second half 2
//       2: "role": the workgroup whose LDS allocation starts at 0 runs its MFMAs at
//          priority 3, the other one at 1
//       3: "late4": priority rises 0,1,2,3 over the four quarters of the MFMA section
//       4: role 1 sleeps ~2000 cycles before its first tile (phase offset), else as 0
//       5: no s_setprio at all
//       6: "early": first half priority 2, second half 1
//       7: LDS refill + next global loads issued after the first 16 MFMAs, no s_setprio
//       8: as 7 with the product's s_setprio
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 128, BK = 16;

__device__ __forceinline__ int lds_chunk_off(int row, int kc) {
  return row * BK + ((2 * kc) ^ (row & 14));
}

template <int P>
__device__ __forceinline__ void setprio() {
  __builtin_amdgcn_s_setprio(P);
}

template <int SCHED>
__global__ __launch_bounds__(256, 2) void k_gemm(const double* __restrict__ A,
                                                 const double* __restrict__ B,
                                                 double* __restrict__ C, int n, int ld, int xcd_chunk,
                                                 unsigned* __restrict__ dbg) {
  __shared__ __attribute__((aligned(16))) double As[2][BM * BK];
  __shared__ __attribute__((aligned(16))) double Bs[2][BN * BK];
  int tile = blockIdx.x;
  tile = (tile & 7) * xcd_chunk + (tile >> 3);
  const int patch = tile >> 6, within = tile & 63;
  const int ppr = n / (8 * BM);
  const int ti = (patch / ppr) * 8 + (within >> 3);
  const int tj = (patch % ppr) * 8 + (within & 7);
  const int row0 = ti * BM, col0 = tj * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, li = lane & 15, lg = lane >> 4;

  const unsigned lds_alloc = __builtin_amdgcn_s_getreg(6 | (31 << 11));  // HW_REG_LDS_ALLOC
  const int role = (lds_alloc & 0xfff) != 0;
  if (dbg != nullptr && tid == 0 && blockIdx.x < 1024) dbg[blockIdx.x] = lds_alloc;

  const double* aptr[4];
  const double* bptr[4];
  int lds_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = tid + 256 * q, r = c >> 3, kc = c & 7;
    aptr[q] = A + (size_t)(row0 + r) * ld + 2 * kc;
    bptr[q] = B + (size_t)(col0 + r) * ld + 2 * kc;
    lds_off[q] = lds_chunk_off(r, kc);
  }
  v4f64 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = (v4f64){0.0, 0.0, 0.0, 0.0};
  const int ktiles = n / BK;
  double2 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ra[q] = *reinterpret_cast<const double2*>(aptr[q] + k0);
      rb[q] = *reinterpret_cast<const double2*>(bptr[q] + k0);
    }
  };
  auto lds_store = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<double2*>(&As[buf][lds_off[q]]) = ra[q];
      *reinterpret_cast<double2*>(&Bs[buf][lds_off[q]]) = rb[q];
    }
  };
  const int arow = (wr * 64 + li) * BK;
  const int brow = (wc * 64 + li) * BK;

  if (SCHED == 4 && role) {
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(8);  // 4 x 8 x 64 cycles
  }
  gload(0);
  lds_store(0);
  if (ktiles > 1) gload(1);
  __syncthreads();

  for (int kt = 0; kt < ktiles; ++kt) {
    const int cur = kt & 1;
    if (SCHED < 7 && kt + 1 < ktiles) {
      lds_store(cur ^ 1);
      if (kt + 2 < ktiles) gload(kt + 2);
    }
    const double* Ac = As[cur];
    const double* Bc = Bs[cur];
    if (SCHED == 0 || SCHED == 4 || SCHED == 8) setprio<1>();
    if (SCHED == 2) {
      if (role) setprio<1>(); else setprio<3>();
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (SCHED == 1) { if (p == 0) setprio<1>(); else setprio<2>(); }
      if (SCHED == 6) { if (p == 0) setprio<2>(); else setprio<1>(); }
      if (SCHED == 3) { if (p == 0) setprio<0>(); else setprio<2>(); }
      const int koff = (2 * (4 * p + lg)) ^ (li & 14);
      double2 a[4], b[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        a[m] = *reinterpret_cast<const double2*>(Ac + arow + m * 16 * BK + koff);
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
        b[nn] = *reinterpret_cast<const double2*>(Bc + brow + nn * 16 * BK + koff);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
          acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[nn].x, acc[m][nn], 0, 0, 0);
      if (SCHED == 3) { if (p == 0) setprio<1>(); else setprio<3>(); }
      if (SCHED >= 7 && p == 0 && kt + 1 < ktiles) {
        // refill in the middle of the MFMA stream: the loads have had 1.25 iterations to land
        __builtin_amdgcn_sched_barrier(0);
        lds_store(cur ^ 1);
        if (kt + 2 < ktiles) gload(kt + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
          acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[nn].y, acc[m][nn], 0, 0, 0);
    }
    if (SCHED != 5 && SCHED != 7) setprio<0>();
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wr * 64 + m * 16 + lg + 4 * r;
        const int col = col0 + wc * 64 + nn * 16 + li;
        C[(size_t)row * ld + col] = acc[m][nn][r];
      }
}

template <int SCHED>
static void run(const double* a, const double* b, double* c, int n, int ld, int reps, unsigned* dbg,
                const std::vector<double>& ha, const std::vector<double>& hb) {
  const int blocks = (n / BM) * (n / BN);
  const int chunk = blocks / 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipMemset(c, 0, (size_t)n * ld * 8);
  for (int it = 0; it < 2; ++it)
    hipLaunchKernelGGL(k_gemm<SCHED>, dim3(blocks), dim3(256), 0, 0, a, b, c, n, ld, chunk, dbg);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < reps; ++it)
    hipLaunchKernelGGL(k_gemm<SCHED>, dim3(blocks), dim3(256), 0, 0, a, b, c, n, ld, chunk,
                       (unsigned*)nullptr);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  double maxerr = 0;
  for (int t = 0; t < 8; ++t) {
    const int i = (t * 977 + 3) % n, j = (t * 3331 + 5) % n;
    double ref = 0;
    for (int k = 0; k < n; ++k) ref += ha[(size_t)i * n + k] * hb[(size_t)j * n + k];
    double got;
    hipMemcpy(&got, c + (size_t)i * ld + j, 8, hipMemcpyDeviceToHost);
    const double e = fabs(got - ref) / fabs(ref);
    if (e > maxerr) maxerr = e;
  }
  printf("sched %d: %.3f ms  %.1f TFLOP/s  (err %.1e, %s)\n", SCHED, ms,
         2.0 * n * n * n / ms / 1e9, maxerr, hipGetErrorString(hipGetLastError()));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::vector<double> ha((size_t)n * n), hb((size_t)n * n);
  unsigned long long s = 88172645463325252ull;
  for (size_t i = 0; i < ha.size(); ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    ha[i] = (double)(s >> 11) / 9007199254740992.0;
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    hb[i] = (double)(s >> 11) / 9007199254740992.0;
  }
  double *a, *b, *c;
  unsigned* dbg;
  const int ld = argc > 3 ? atoi(argv[3]) : n + 16;  // the product pads power-of-two strides
  hipMalloc(&a, (size_t)n * ld * 8);
  hipMalloc(&b, (size_t)n * ld * 8);
  hipMalloc(&c, (size_t)n * ld * 8);
  hipMalloc(&dbg, 1024 * 4);
  hipMemset(dbg, 0, 1024 * 4);
  hipMemcpy2D(a, (size_t)ld * 8, ha.data(), (size_t)n * 8, (size_t)n * 8, n, hipMemcpyHostToDevice);
  hipMemcpy2D(b, (size_t)ld * 8, hb.data(), (size_t)n * 8, (size_t)n * 8, n, hipMemcpyHostToDevice);
  printf("n = %d, ld = %d\n", n, ld);
  run<0>(a, b, c, n, ld, reps, dbg, ha, hb);
  std::vector<unsigned> hd(1024);
  hipMemcpy(hd.data(), dbg, 1024 * 4, hipMemcpyDeviceToHost);
  printf("LDS_ALLOC of workgroups 0..7, 256..263, 512..519:\n");
  for (int base : {0, 256, 512}) {
    for (int i = 0; i < 8; ++i) printf(" %08x", hd[base + i]);
    printf("\n");
  }
  run<1>(a, b, c, n, ld, reps, nullptr, ha, hb);
  run<2>(a, b, c, n, ld, reps, nullptr, ha, hb);
  run<3>(a, b, c, n, ld, reps, nullptr, ha, hb);
  run<4>(a, b, c, n, ld, reps, nullptr, ha, hb);
  run<5>(a, b, c, n, ld, reps, nullptr, ha, hb);
  run<6>(a, b, c, n, ld, reps, nullptr, ha, hb);
  run<7>(a, b, c, n, ld, reps, nullptr, ha, hb);
  run<8>(a, b, c, n, ld, reps, nullptr, ha, hb);
  run<0>(a, b, c, n, ld, reps, nullptr, ha, hb);
  return 0;
}
