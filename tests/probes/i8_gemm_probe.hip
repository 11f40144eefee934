// Probe (not on the product path): what bounds k_gemm_i8_sym (csrc/diffuse_free.hip)?  The same
// kernel with one of its three engines removed, on random digits, n = 8192.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spectralcluster_amd/csrc -o /tmp/i8_gemm_probe tests/probes/i8_gemm_probe.hip
//   /tmp/i8_gemm_probe [n] [reps]
// PROBE 0: the product's kernel; 1: no LDS-DMA inside the K loop (MFMA + fragment reads only);
//       2: no MFMA (DMA + fragment reads); 3: no fragment reads inside the loop (DMA + MFMA).
#include "diffuse_free.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace sc;

static unsigned long long* g_clk = nullptr;
static double g_cycles = 0, g_ticks = 0;

template <int PROBE>
static float run(const signed char* Q, int n, const int2* tilemap, float* T32, unsigned* M, int reps,
                 int xcd) {
  const int nt = (n + kI8Tile - 1) / kI8Tile, tiles = nt * (nt + 1) / 2, Kp = free_k_padded(n);
  const int lds = kI8Buffers * kI8StageBytes;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_i8_sym<PROBE>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int xcd_chunk = (xcd && tiles % 8 == 0 && tiles >= 512) ? tiles / 8 : 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_gemm_i8_sym<PROBE>, dim3(tiles), dim3(kI8Threads), lds, 0, Q,
                       (size_t)2 * Kp, Kp / 64, tilemap, xcd_chunk, T32, nt, n, M, g_clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(2 * tiles);
  hipMemcpy(h.data(), g_clk, h.size() * 8, hipMemcpyDeviceToHost);
  g_cycles = g_ticks = 0;
  for (int t = 0; t < tiles; ++t) { g_cycles += h[2 * t]; g_ticks += h[2 * t + 1]; }
  g_cycles /= tiles;
  g_ticks /= tiles;
  return best;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const int nt = (n + kI8Tile - 1) / kI8Tile, tiles = nt * (nt + 1) / 2;
  std::vector<signed char> hq(free_q_bytes(n));
  unsigned x = 12345u;
  for (auto& b : hq) { x = x * 1664525u + 1013904223u; b = (signed char)(x >> 24); }
  std::vector<int2> map;

  {
    const int np = (nt + 7) / 8;
    for (int pi = 0; pi < np; ++pi)
      for (int pj = pi; pj < np; ++pj)
        for (int ti = pi * 8; ti < std::min(nt, pi * 8 + 8); ++ti)
          for (int tj = std::max(ti, pj * 8); tj < std::min(nt, pj * 8 + 8); ++tj)
            map.push_back(make_int2(ti, tj));
  }
  signed char* Q;
  int2* tm;
  float* T32;
  unsigned* M;
  hipMalloc(&Q, hq.size());
  hipMalloc(&tm, map.size() * sizeof(int2));
  hipMalloc(&T32, free_t32_bytes(n));
  hipMalloc(&M, n * sizeof(unsigned));
  hipMemcpy(Q, hq.data(), hq.size(), hipMemcpyHostToDevice);
  hipMemcpy(tm, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice);
  hipMemset(M, 0, n * sizeof(unsigned));
  hipMalloc(&g_clk, 2 * tiles * sizeof(unsigned long long));
  const double ops = 4.0 * tiles * 2.0 * 128 * 128 * free_k_padded(n);
  auto clk = [&](const char* what) {
    printf("    %-22s K loop: %.0f shader cycles per tile (MFMA-bound %d), %.1f us, effective "
           "clock %.0f MHz\n", what, g_cycles, (free_k_padded(n) / 64) * 1024, g_ticks / 100.0,
           g_cycles / (g_ticks / 100.0));
  };
  const float t0 = run<0>(Q, n, tm, T32, M, reps, 1);
  clk("product");
  const float t0n = run<0>(Q, n, tm, T32, M, reps, 0);
  const float t1 = run<1>(Q, n, tm, T32, M, reps, 1);
  clk("no DMA");
  const float t2 = run<2>(Q, n, tm, T32, M, reps, 1);
  clk("no MFMA");
  const float t3 = run<3>(Q, n, tm, T32, M, reps, 1);
  clk("no reads");
  // the same on mostly-small digits (what a thresholded affinity looks like: 7/8 of the high
  // digits are 0 or 1)
  for (size_t i = 0; i < hq.size(); ++i)
    if (((i >> 6) & 1) == 0 && (i * 2654435761u >> 29) != 0) hq[i] = (signed char)(hq[i] & 1);
  hipMemcpy(Q, hq.data(), hq.size(), hipMemcpyHostToDevice);
  const float t0s = run<0>(Q, n, tm, T32, M, reps, 1);
  clk("product, small digits");
  printf("  product kernel, 7/8 of the high digits in {0, 1}: %.3f ms\n", t0s);
  printf("n=%d tiles=%d  (MFMA-bound at 2.4 GHz: %.3f ms with the 9-round tail, %.3f without)\n", n,
         tiles, ((tiles + 255) / 256) * (free_k_padded(n) / 64) * 1024.0 / 2.4e6,
         tiles / 256.0 * (free_k_padded(n) / 64) * 1024.0 / 2.4e6);
  printf("  product kernel          %.3f ms  %.0f TOP/s\n", t0, ops / t0 / 1e9);
  printf("  ... identity tile order %.3f ms\n", t0n);
  printf("  no DMA in the loop      %.3f ms\n", t1);
  printf("  no MFMA                 %.3f ms\n", t2);
  printf("  no fragment reads       %.3f ms\n", t3);
  return 0;
}
