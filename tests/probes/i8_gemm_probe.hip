// Probe (not on the product path): what bounds k_gemm_i8_sym (csrc/diffuse_free.hip)?  The same
// kernel with one of its three engines removed, on random digits, n = 8192.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spectralcluster_amd/csrc -o /tmp/i8_gemm_probe tests/probes/i8_gemm_probe.hip
//   /tmp/i8_gemm_probe [n] [reps]
// PROBE 0: the product's kernel; 1: no LDS-DMA inside the K loop (MFMA + fragment reads only);
//       2: no MFMA (DMA + fragment reads); 3: no fragment reads inside the loop (DMA + MFMA).
#include "diffuse_free.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace sc {
// ---- experiment (not in the library): the product as TWO 4-wave workgroups per CU.
// Result (profiles/r04q_i8_gemm_forms.txt): the barrier no longer costs anything and without
// its DMA the loop is exactly MFMA-bound, but with one stage of prefetch distance (two 32 KB
// buffers per workgroup are all that fits twice in 160 KB) the DMA adds 37 % -- 0.99 ms against
// 1.10 on random digits, 0.82-0.84 against 0.79-0.80 on a real thresholded affinity.  Not kept.
// tests/probes/i8_gemm_probe.hip: the s_barrier of the 8-wave form costs 216 of a stage's 1317
// cycles (every wave of the CU waits at it together).  Here a workgroup is 4 waves (one per
// SIMD), wave tile 64 x 64 = 2 x 2 MFMA blocks x three accumulators = 192 VGPRs, two workgroups
// per CU with a barrier each: while one stands at its barrier the other one's waves own the
// matrix pipes.  Two 32 KB stage buffers per workgroup (the DMA of stage st + 1 is issued right
// behind the barrier of stage st and has a whole stage to land), fragments single-buffered in
// registers (the partner workgroup covers their LDS latency), same LDS image and swizzle.
constexpr int kI8Threads4 = 256;
template <int PROBE>
__global__ __launch_bounds__(kI8Threads4, 2) void k_gemm_i8_sym4(
    const signed char* __restrict__ Q, size_t pitch, int nstages, const int2* __restrict__ tilemap,
    int xcd_chunk, float* __restrict__ T32, int nt, int n, unsigned* __restrict__ M,
    unsigned long long* __restrict__ probe_clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned long long c_begin = probe_clk ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long t_begin = probe_clk ? wall_clock64() : 0ull;
  int tile = blockIdx.x;
  if (xcd_chunk > 0) tile = (tile & 7) * xcd_chunk + (tile >> 3);
  const int2 tij = tilemap[tile];
  const int I = tij.x, J = tij.y;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // DMA: instruction q (0..7) of wave w fills units [q * 256 + w * 64, + 64) of [A tile | B tile]
  const int sr = 8 * w + (lane >> 3);
  const int cx = (lane & 7) ^ ((4 * w + (lane >> 4)) & 7);
  const signed char* gA = Q + (size_t)(I * kI8Tile + sr) * pitch + 16 * cx;
  const signed char* gB = Q + (size_t)(J * kI8Tile + sr) * pitch + 16 * cx;
  const size_t row32 = (size_t)32 * pitch;
  auto issue = [&](int stage) {
    if (PROBE == 1 && stage > 0) return;
    unsigned char* base = lds + (stage & 1) * kI8StageBytes + w * 1024;
    const size_t off = (size_t)stage * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(gA + q * row32 + off, base + q * 4096);
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(gB + q * row32 + off, base + 16384 + q * 4096);
  };
  const int rr = lane & 31, g = lane >> 5;
  const int y = g ^ ((rr >> 1) & 7);
  const int wr = w >> 1, wc = w & 1;
  const int aoff = (64 * wr + rr) * 128;
  const int boff = 16384 + (64 * wc + rr) * 128;
  int coff[2][2];
#pragma unroll
  for (int dg = 0; dg < 2; ++dg)
#pragma unroll
    for (int s = 0; s < 2; ++s) coff[dg][s] = 16 * ((4 * dg + 2 * s) ^ y);
  v16i hh[2][2], mid[2][2], ll[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hh[rb][cb][r] = 0;
        mid[rb][cb][r] = 0;
        ll[rb][cb][r] = 0;
      }
  issue(0);
  for (int st = 0; st < nstages; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of stage st landed
    if (PROBE != 4) __builtin_amdgcn_s_barrier();     // ... everybody's did; stage st - 1 is read out
    if (st + 1 < nstages) issue(st + 1);
    const unsigned char* sb = lds + (st & 1) * kI8StageBytes;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      v4i ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        ah[b] = *reinterpret_cast<const v4i*>(sb + aoff + b * 4096 + coff[0][s]);
        al[b] = *reinterpret_cast<const v4i*>(sb + aoff + b * 4096 + coff[1][s]);
        bh[b] = *reinterpret_cast<const v4i*>(sb + boff + b * 4096 + coff[0][s]);
        bl[b] = *reinterpret_cast<const v4i*>(sb + boff + b * 4096 + coff[1][s]);
      }
      if (PROBE == 2) {
        asm volatile("" ::"v"(ah[0]), "v"(ah[1]), "v"(al[0]), "v"(al[1]), "v"(bh[0]), "v"(bh[1]),
                     "v"(bl[0]), "v"(bl[1]));
        continue;
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          hh[rb][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah[rb], bh[cb], hh[rb][cb], 0, 0, 0);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          mid[rb][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah[rb], bl[cb], mid[rb][cb], 0, 0, 0);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          ll[rb][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al[rb], bl[cb], ll[rb][cb], 0, 0, 0);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          mid[rb][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al[rb], bh[cb], mid[rb][cb], 0, 0, 0);
    }
  }
  if (probe_clk != nullptr && threadIdx.x == 0) {
    probe_clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c_begin;
    probe_clk[2 * blockIdx.x + 1] = wall_clock64() - t_begin;
  }
  // ---- epilogue (as in the 8-wave form)
  float* out = T32 + (size_t)tile_to_slot(I, J, nt) * (kI8Tile * kI8Tile);
  float rowm[2][16], colm[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) rowm[rb][r] = -INFINITY;
  colm[0] = colm[1] = -INFINITY;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int grow0 = I * kI8Tile + 64 * wr + 32 * rb + 4 * g;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = 64 * wc + 32 * cb + rr;
      const bool col_ok = J * kI8Tile + col < n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 64 * wr + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * g;
        const double t = (double)hh[rb][cb][r] * 65536.0 + (double)mid[rb][cb][r] * 256.0 +
                         (double)ll[rb][cb][r];
        const float tf = (float)t;
        out[row * kI8Tile + col] = tf;
        if (col_ok) rowm[rb][r] = fmaxf(rowm[rb][r], tf);
        if (grow0 + (r & 3) + 8 * (r >> 2) < n) colm[cb] = fmaxf(colm[cb], tf);
      }
    }
  }
  __syncthreads();
  float* prow = reinterpret_cast<float*>(lds);  // rows [wc][128], columns [wr][128]
  float* pcol = prow + 2 * 128;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = rowm[rb][r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
      if (rr == 0) prow[wc * 128 + 64 * wr + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * g] = v;
    }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const float v = fmaxf(colm[cb], __shfl_xor(colm[cb], 32));
    if (g == 0) pcol[wr * 128 + 64 * wc + 32 * cb + rr] = v;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int row = I * kI8Tile + threadIdx.x;
    const float v = fmaxf(prow[threadIdx.x], prow[128 + threadIdx.x]);
    if (row < n && v > -INFINITY) atomicMax(&M[row], ordered_bits(v));
  } else if (I != J) {
    const int c = threadIdx.x - 128;
    const int row = J * kI8Tile + c;
    const float v = fmaxf(pcol[c], pcol[128 + c]);
    if (row < n && v > -INFINITY) atomicMax(&M[row], ordered_bits(v));
  }
}

}  // namespace sc

using namespace sc;

static unsigned long long* g_clk = nullptr;
static double g_cycles = 0, g_ticks = 0;

template <int PROBE>
static float run4(const signed char* Q, int n, const int2* tilemap, float* T32, unsigned* M, int reps,
                  int xcd) {
  const int nt = (n + kI8Tile - 1) / kI8Tile, tiles = nt * (nt + 1) / 2, Kp = free_k_padded(n);
  const int lds = 2 * kI8StageBytes;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_i8_sym4<PROBE>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int xcd_chunk = (xcd && tiles % 8 == 0 && tiles >= 512) ? tiles / 8 : 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_gemm_i8_sym4<PROBE>, dim3(tiles), dim3(kI8Threads4), lds, 0, Q,
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
                       (size_t)2 * Kp, Kp / 64, tilemap, xcd_chunk, T32, nt, n, M, g_clk,
                       I8Split{tiles, 1, nullptr});
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
  const float t4 = run<4>(Q, n, tm, T32, M, reps, 1);
  clk("no barrier");
  printf("  no s_barrier in the loop  %.3f ms (wrong results: timing only)\n", t4);
  {  // the two-workgroups-per-CU form
    std::vector<float> ref(16384), got(16384);
    run<0>(Q, n, tm, T32, M, 1, 1);
    hipMemcpy(ref.data(), T32 + (size_t)777 * 16384, 16384 * 4, hipMemcpyDeviceToHost);
    hipMemset(T32, 0, free_t32_bytes(n));
    const float u0 = run4<0>(Q, n, tm, T32, M, reps, 1);
    clk("4-wave x2 product");
    hipMemcpy(got.data(), T32 + (size_t)777 * 16384, 16384 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16384; ++i) bad += ref[i] != got[i];
    const float u1 = run4<1>(Q, n, tm, T32, M, reps, 1);
    clk("4-wave x2 no DMA");
    const float u2 = run4<2>(Q, n, tm, T32, M, reps, 1);
    clk("4-wave x2 no MFMA");
    const float u4 = run4<4>(Q, n, tm, T32, M, reps, 1);
    clk("4-wave x2 no barrier");
    printf("  4-wave x 2-per-CU form: %.3f ms (tile 777 differs from the 8-wave form in %d entries); "
           "no DMA %.3f, no MFMA %.3f, no barrier %.3f\n", u0, bad, u1, u2, u4);
  }
  {  // the library's launcher with and without the split-K tail: same bits, less time?
    int r = 0, f = 1;
    free_i8_split_plan(n, &r, &f);
    int* ws = nullptr;
    unsigned* cnt = nullptr;
    hipMalloc(&ws, std::max<size_t>(free_i8_split_bytes(n), 16));
    hipMalloc(&cnt, 256 * sizeof(unsigned));
    hipMemset(cnt, 0, 256 * sizeof(unsigned));
    const size_t words = free_t32_bytes(n) / 4;
    std::vector<float> a(words), b(words);
    std::vector<unsigned> ma(n), mb(n);
    auto timed = [&](int* w, unsigned* c) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float best = 1e9f;
      for (int it = 0; it < reps + 1; ++it) {
        hipMemset(M, 0, n * sizeof(unsigned));
        hipEventRecord(e0, 0);
        launch_gemm_i8_sym(0, Q, n, tm, T32, M, w);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
      }
      return best;
    };
    hipMemset(T32, 0, free_t32_bytes(n));
    const float plain = timed(nullptr, nullptr);
    hipMemcpy(a.data(), T32, words * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ma.data(), M, n * 4, hipMemcpyDeviceToHost);
    hipMemset(T32, 0, free_t32_bytes(n));
    const float cut = timed(ws, cnt);
    hipMemcpy(b.data(), T32, words * 4, hipMemcpyDeviceToHost);
    hipMemcpy(mb.data(), M, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, badm = 0;
    for (size_t i = 0; i < words; ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
    for (int i = 0; i < n; ++i) badm += ma[i] != mb[i];
    std::vector<unsigned> hc(256);
    hipMemcpy(hc.data(), cnt, 256 * 4, hipMemcpyDeviceToHost);
    unsigned left = 0;
    for (unsigned v : hc) left += v;
    printf("  split-K tail: %d tiles x %d parts: %.3f ms against %.3f without; T32 differs in %zu "
           "words, M in %zu rows, counters left %u\n", r, f, cut, plain, bad, badm, left);
  }
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
