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

// ---- experiment: 256 x 128 workgroup tiles.  8 waves, wave tile 64 x 64 (2 x 2 MFMA blocks x
// three accumulators = 192 VGPRs), stage = 256 + 128 rows x 128 B = 48 KB, three stage buffers
// filled two stages ahead.  Per MFMA: 0.75 x the DMA bytes, 0.67 x the fragment reads and half the
// barriers of the 128 x 128 form.  Fragments are single-buffered and reloaded on a rolling
// schedule: each k-step runs its four digit products in an order that frees one operand group
// after the second product, one after the third and two at the end, and each group is reloaded
// the moment it is free -- so the next step's first product always finds its two operands
// loaded at least four MFMAs ago, and the two groups requested at the step boundary have that
// product's four MFMAs (twice that with the SIMD's other wave) to arrive.
constexpr int kI8WideStage = 49152;
template <int PROBE>
__global__ __launch_bounds__(kI8Threads) void k_gemm_i8_wide(
    const signed char* __restrict__ Q, size_t pitch, int nstages, const int2* __restrict__ tilemap,
    int xcd_chunk, float* __restrict__ T32, int nt, int n, unsigned* __restrict__ M,
    unsigned long long* __restrict__ probe_clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned long long c_begin = probe_clk ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long t_begin = probe_clk ? wall_clock64() : 0ull;
  int tile = blockIdx.x;
  if (xcd_chunk > 0) tile = (tile & 7) * xcd_chunk + (tile >> 3);
  const int2 tij = tilemap[tile];
  const int I2 = tij.x, J = tij.y;  // rows [256 I2, + 256), columns [128 J, + 128), J >= 2 I2
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // DMA: piece q of wave w = 64 units (16 B) = rows 64 q + 8 w + (lane >> 3) of the A part
  // (q < 4) or rows 64 (q - 4) + ... of the B part; unit = row * 8 + stored chunk
  const int sr = 8 * w + (lane >> 3);
  const int cx = (lane & 7) ^ ((4 * w + (lane >> 4)) & 7);
  const int last_row = nt * kI8Tile - 1;
  const signed char* gsrc[6];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int row = I2 * 256 + 64 * q + sr;
    row = row > last_row ? last_row : row;  // (an odd tile count: the lower half does not exist)
    gsrc[q] = Q + (size_t)row * pitch + 16 * cx;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) gsrc[4 + q] = Q + (size_t)(J * kI8Tile + 64 * q + sr) * pitch + 16 * cx;
  auto piece = [&](int stage, int q) {
    if (PROBE == 1 && stage > 1) return;
    glds16(gsrc[q] + (size_t)stage * 128, lds + (stage % 3) * kI8WideStage + w * 1024 + q * 8192);
  };
  auto issue = [&](int stage) {
#pragma unroll
    for (int q = 0; q < 6; ++q) piece(stage, q);
  };
  const int rr = lane & 31, g = lane >> 5;
  const int y = g ^ ((rr >> 1) & 7);
  const int wr = w >> 1, wc = w & 1;
  const int aoff = (64 * wr + rr) * 128;
  const int boff = 32768 + (64 * wc + rr) * 128;
  int coff[2][2];
#pragma unroll
  for (int dg = 0; dg < 2; ++dg)
#pragma unroll
    for (int s = 0; s < 2; ++s) coff[dg][s] = 16 * ((4 * dg + 2 * s) ^ y);
  v4i ah[2], al[2], bh[2], bl[2];
  auto rd = [&](v4i* f, const unsigned char* sb, int off, int dg, int s) {
    if (PROBE == 3 && sb != lds) return;
    f[0] = *reinterpret_cast<const v4i*>(sb + off + coff[dg][s]);
    f[1] = *reinterpret_cast<const v4i*>(sb + off + 4096 + coff[dg][s]);
  };
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
  auto prod = [&](v16i (&acc)[2][2], const v4i* a, const v4i* b) {
    if (PROBE == 2) {
      asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
      return;
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        acc[rb][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rb], b[cb], acc[rb][cb], 0, 0, 0);
  };
#define SB __builtin_amdgcn_sched_barrier(0)
#define LGKM0 __builtin_amdgcn_s_waitcnt(0xc07f)
  issue(0);
  if (nstages > 1) issue(1);
  if (nstages > 1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  rd(ah, lds, aoff, 0, 0);
  rd(bh, lds, boff, 0, 0);
  SB; LGKM0;  // (the loop is entered in the state its own back edge leaves: ah, bh there, al, bl under way)
  rd(al, lds, aoff, 1, 0);
  rd(bl, lds, boff, 1, 0);
  SB;
  for (int st = 0; st < nstages; ++st) {
    const unsigned char* sb = lds + (st % 3) * kI8WideStage;
    const unsigned char* sbn = lds + ((st + 1 < nstages ? st + 1 : st) % 3) * kI8WideStage;
    const bool more = st + 2 < nstages;
    // (hipcc waits for ALL outstanding LDS reads, lgkmcnt(0), wherever one of them is needed --
    //  never for "all but the last four".  So the reads are placed such that every wait finds
    //  only reads that are at least four MFMAs old: LGKM0 drains the two groups requested
    //  during the step before the boundary pair is requested, and the compiler's own wait in
    //  front of the second product drains that pair.)
    // ---- k-step 0: hh | mid (ah free) | ll (bl free) | mid (al, bh free)
    // (one piece in front of each product instead of six in a row here: 351 k cycles per tile
    //  against 341 k -- and 188 k against 164 k when the 128 x 128 form was given the same)
    if (more) issue(st + 2);
    SB; prod(hh, ah, bh);
    SB; prod(mid, ah, bl);
    SB; rd(ah, sb, aoff, 0, 1);
    SB; prod(ll, al, bl);
    SB; rd(bl, sb, boff, 1, 1);
    SB; prod(mid, al, bh);
    SB; LGKM0; rd(al, sb, aoff, 1, 1); rd(bh, sb, boff, 0, 1);
    // ---- k-step 1: mid (ah, bl) | hh (ah free) | mid (bh free) | ll (al, bl free)
    SB; prod(mid, ah, bl);
    SB; prod(hh, ah, bh);
    SB;
    // everybody's pieces of stage st + 1 have landed and nobody reads stage st - 1 any more
    if (more && PROBE != 1)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PROBE != 4) __builtin_amdgcn_s_barrier();
    SB; rd(ah, sbn, aoff, 0, 0);
    SB; prod(mid, al, bh);
    SB; rd(bh, sbn, boff, 0, 0);
    SB; prod(ll, al, bl);
    SB; LGKM0; rd(al, sbn, aoff, 1, 0); rd(bl, sbn, boff, 1, 0);
    SB;
  }
#undef LGKM0
#undef SB
  if (probe_clk != nullptr && threadIdx.x == 0) {
    probe_clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c_begin;
    probe_clk[2 * blockIdx.x + 1] = wall_clock64() - t_begin;
  }
  // ---- epilogue: the wave's 64 rows lie in ONE of the two 128-row tiles, Is = 2 I2 + (wr >> 1)
  const int Is = 2 * I2 + (wr >> 1);
  const bool stored = Is <= J && Is < nt;  // (the other half of a diagonal tile is T's lower triangle)
  float* out = T32 + (size_t)tile_to_slot(stored ? Is : 0, stored ? J : 0, nt) * (kI8Tile * kI8Tile);
  float rowm[2][16], colm[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) rowm[rb][r] = -INFINITY;
  colm[0] = colm[1] = -INFINITY;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int lrow0 = 64 * (wr & 1) + 32 * rb + 4 * g;  // row inside the 128-row tile
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = 64 * wc + 32 * cb + rr;
      const bool col_ok = J * kI8Tile + col < n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = lrow0 + (r & 3) + 8 * (r >> 2);
        const double t = (double)hh[rb][cb][r] * 65536.0 + (double)mid[rb][cb][r] * 256.0 +
                         (double)ll[rb][cb][r];
        const float tf = (float)t;
        if (stored) out[lrow * kI8Tile + col] = tf;
        if (col_ok) rowm[rb][r] = fmaxf(rowm[rb][r], tf);
        if (Is * kI8Tile + lrow < n) colm[cb] = fmaxf(colm[cb], tf);
      }
    }
  }
  __syncthreads();
  float* prow = reinterpret_cast<float*>(lds);  // rows [wc][256], columns [wr][128]
  float* pcol = prow + 2 * 256;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = rowm[rb][r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
      if (rr == 0) prow[wc * 256 + 64 * wr + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * g] = v;
    }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const float v = fmaxf(colm[cb], __shfl_xor(colm[cb], 32));
    if (g == 0) pcol[wr * 128 + 64 * wc + 32 * cb + rr] = v;
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int row = I2 * 256 + threadIdx.x;
    const float v = fmaxf(prow[threadIdx.x], prow[256 + threadIdx.x]);
    if (row < n && v > -INFINITY) atomicMax(&M[row], ordered_bits(v));
  } else if (threadIdx.x < 384) {
    const int c = threadIdx.x - 256;
    const int row = J * kI8Tile + c;
    const float v = fmaxf(fmaxf(pcol[c], pcol[128 + c]), fmaxf(pcol[256 + c], pcol[384 + c]));
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

template <int PROBE>
static float runW(const signed char* Q, int n, const int2* wmap, int wtiles, float* T32, unsigned* M,
                  int reps) {
  const int nt = (n + kI8Tile - 1) / kI8Tile, Kp = free_k_padded(n);
  const int lds = 3 * kI8WideStage;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_i8_wide<PROBE>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int xcd_chunk = (wtiles % 8 == 0 && wtiles >= 512) ? wtiles / 8 : 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_gemm_i8_wide<PROBE>, dim3(wtiles), dim3(kI8Threads), lds, 0, Q,
                       (size_t)2 * Kp, Kp / 64, wmap, xcd_chunk, T32, nt, n, M, g_clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(2 * wtiles);
  hipMemcpy(h.data(), g_clk, h.size() * 8, hipMemcpyDeviceToHost);
  g_cycles = g_ticks = 0;
  for (int t = 0; t < wtiles; ++t) { g_cycles += h[2 * t]; g_ticks += h[2 * t + 1]; }
  g_cycles /= wtiles;
  g_ticks /= wtiles;
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
  {  // every tile = tile (0, 0): all DMA traffic hits L2 (results meaningless, timing only)
    std::vector<int2> zmap(map.size(), make_int2(0, 0));
    int2* zm;
    hipMalloc(&zm, zmap.size() * sizeof(int2));
    hipMemcpy(zm, zmap.data(), zmap.size() * sizeof(int2), hipMemcpyHostToDevice);
    const float z0 = run<0>(Q, n, zm, T32, M, reps, 1);
    clk("all tiles = (0, 0)");
    // ... and with the patch order but only 16 distinct panels (tile indices mod 16)
    for (auto& t : zmap) t = make_int2(0, 0);
    for (size_t i = 0; i < map.size(); ++i) zmap[i] = make_int2(map[i].x % 16, map[i].y % 16);
    hipMemcpy(zm, zmap.data(), zmap.size() * sizeof(int2), hipMemcpyHostToDevice);
    const float z1 = run<0>(Q, n, zm, T32, M, reps, 1);
    clk("tile indices mod 16");
    printf("  DMA from L2 only: %.3f ms; 16 distinct panels (32 MB of digits: MALL): %.3f ms\n", z0, z1);
    hipFree(zm);
  }
  {  // 256 x 128 workgroup tiles
    std::vector<int2> wmap;
    const int nt2 = (nt + 1) / 2;
    const int np = (nt2 + 3) / 4;  // patches of 4 x 8 wide tiles (1024 x 1024 entries of T)
    for (int pi = 0; pi < np; ++pi)
      for (int pj = pi; pj < (nt + 7) / 8 + 0; ++pj)
        for (int ti = pi * 4; ti < std::min(nt2, pi * 4 + 4); ++ti)
          for (int tj = std::max(2 * ti, pj * 8); tj < std::min(nt, pj * 8 + 8); ++tj)
            wmap.push_back(make_int2(ti, tj));
    int2* wm;
    hipMalloc(&wm, wmap.size() * sizeof(int2));
    hipMemcpy(wm, wmap.data(), wmap.size() * sizeof(int2), hipMemcpyHostToDevice);
    const size_t words = free_t32_bytes(n) / 4;
    std::vector<float> a(words), b(words);
    std::vector<unsigned> ma(n), mb(n);
    hipMemset(M, 0, n * sizeof(unsigned));
    hipMemset(T32, 0, free_t32_bytes(n));
    launch_gemm_i8_sym(0, Q, n, tm, T32, M, nullptr);
    hipMemcpy(a.data(), T32, words * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ma.data(), M, n * 4, hipMemcpyDeviceToHost);
    hipMemset(M, 0, n * sizeof(unsigned));
    hipMemset(T32, 0, free_t32_bytes(n));
    runW<0>(Q, n, wm, (int)wmap.size(), T32, M, 0);
    hipMemcpy(b.data(), T32, words * 4, hipMemcpyDeviceToHost);
    hipMemcpy(mb.data(), M, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, badm = 0;
    for (size_t i = 0; i < words; ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
    for (int i = 0; i < n; ++i) badm += ma[i] != mb[i];
    const float w0 = runW<0>(Q, n, wm, (int)wmap.size(), T32, M, reps);
    clk("wide product");
    const float w1 = runW<1>(Q, n, wm, (int)wmap.size(), T32, M, reps);
    clk("wide no DMA");
    const float w2 = runW<2>(Q, n, wm, (int)wmap.size(), T32, M, reps);
    clk("wide no MFMA");
    const float w3 = runW<3>(Q, n, wm, (int)wmap.size(), T32, M, reps);
    clk("wide no reads");
    const float w4 = runW<4>(Q, n, wm, (int)wmap.size(), T32, M, reps);
    clk("wide no barrier");
    printf("  256 x 128 tiles (%zu of them; per-tile MFMA-bound is TWICE the figure printed): %.3f ms; "
           "T32 differs in %zu words, M in %zu rows; no DMA %.3f, no MFMA %.3f, no reads %.3f, "
           "no barrier %.3f\n", wmap.size(), w0, bad, badm, w1, w2, w3, w4);
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
