// Microbenchmark (not on the product path): the fp64 MFMA stream of csrc/gemm_f64.hip fed from
// LDS with RANDOM data (8 ds_read_b128 -> 32 MFMAs, two workgroups per CU), accumulators in
// arch VGPRs vs AccVGPRs.  Reports the effective shader clock (s_memtime / s_memrealtime):
// does the accumulator file change the power the stream draws?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/bin/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define MFV(i, a, b) "v_mfma_f64_16x16x4_f64 v[" #i ":" #i "+7], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #i ":" #i "+7]\n"
#define MFA(i, a, b) "v_mfma_f64_16x16x4_f64 a[" #i ":" #i "+7], v[" #a ":" #a "+1], v[" #b ":" #b "+1], a[" #i ":" #i "+7]\n"
// A fragments v[160:175] (4 x b128), B fragments v[176:191]; x halves then y halves
#define GROUP(M, o)                                                                        \
  M(0, 160 + o, 176 + o) M(8, 160 + o, 180 + o) M(16, 160 + o, 184 + o) M(24, 160 + o, 188 + o)   \
  M(32, 164 + o, 176 + o) M(40, 164 + o, 180 + o) M(48, 164 + o, 184 + o) M(56, 164 + o, 188 + o) \
  M(64, 168 + o, 176 + o) M(72, 168 + o, 180 + o) M(80, 168 + o, 184 + o) M(88, 168 + o, 188 + o) \
  M(96, 172 + o, 176 + o) M(104, 172 + o, 180 + o) M(112, 172 + o, 184 + o) M(120, 172 + o, 188 + o)
#define GROUPX(M) GROUP(M, 0)
#define GROUPY(M) GROUP(M, 2)
#define READS                                                                              \
  "ds_read_b128 v[160:163], %1\n ds_read_b128 v[164:167], %1 offset:2048\n"                  \
  "ds_read_b128 v[168:171], %1 offset:4096\n ds_read_b128 v[172:175], %1 offset:6144\n"       \
  "ds_read_b128 v[176:179], %1 offset:8192\n ds_read_b128 v[180:183], %1 offset:10240\n"      \
  "ds_read_b128 v[184:187], %1 offset:12288\n ds_read_b128 v[188:191], %1 offset:14336\n"
#define LOOP(M)                                                                            \
  "s_mov_b32 s20, %0\n"                                                                     \
  "1:\n" READS "s_waitcnt lgkmcnt(0)\n" GROUPX(M) GROUPY(M)                                 \
  "v_xor_b32 %1, 0x4000, %1\n"                                                              \
  "s_sub_u32 s20, s20, 1\n"                                                                 \
  "s_cmp_lg_u32 s20, 0\n"                                                                   \
  "s_cbranch_scc1 1b\n"                                                                     \
  "s_nop 15\n"
#define CL(p)                                                                            \
  p "0", p "1", p "2", p "3", p "4", p "5", p "6", p "7", p "8", p "9", p "10", p "11",  \
  p "12", p "13", p "14", p "15", p "16", p "17", p "18", p "19", p "20", p "21", p "22", \
  p "23", p "24", p "25", p "26", p "27", p "28", p "29", p "30", p "31", p "32", p "33", \
  p "34", p "35", p "36", p "37", p "38", p "39", p "40", p "41", p "42", p "43", p "44", \
  p "45", p "46", p "47", p "48", p "49", p "50", p "51", p "52", p "53", p "54", p "55", \
  p "56", p "57", p "58", p "59", p "60", p "61", p "62", p "63", p "64", p "65", p "66", \
  p "67", p "68", p "69", p "70", p "71", p "72", p "73", p "74", p "75", p "76", p "77", \
  p "78", p "79", p "80", p "81", p "82", p "83", p "84", p "85", p "86", p "87", p "88", \
  p "89", p "90", p "91", p "92", p "93", p "94", p "95", p "96", p "97", p "98", p "99", \
  p "100", p "101", p "102", p "103", p "104", p "105", p "106", p "107", p "108",        \
  p "109", p "110", p "111", p "112", p "113", p "114", p "115", p "116", p "117",        \
  p "118", p "119", p "120", p "121", p "122", p "123", p "124", p "125", p "126", p "127"
#define FRAGS                                                                              \
  "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170",   \
  "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181",   \
  "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191"

// MODE 0: acc in arch VGPRs, 1: acc in AccVGPRs
// DATA 0: constant, 1: random in [-0.5, 0.5), 2: random in [0, 1)
// LOADS 0: none, 1: 8 global_load_dwordx4 per 32 MFMAs from a 2 MB window (L2 hits),
//       2: the same from a 1 GB stream (HBM)
template <int MODE, int DATA, int LOADS>
__global__ __launch_bounds__(256, 2) void k_probe(double* out, int iters, double* dbg,
                                                  const double* src, unsigned mask) {
  __shared__ __attribute__((aligned(16))) double lds[4096];  // 32 KB
  unsigned long long h = 0x9E3779B97F4A7C15ull * (blockIdx.x * 256 + threadIdx.x + 1);
  for (int i = threadIdx.x; i < 4096; i += 256) {
    h ^= h << 13; h ^= h >> 7; h ^= h << 17;
    const double u = __longlong_as_double((long long)((h >> 12) | 0x3FF0000000000000ull));
    lds[i] = DATA == 0 ? 0.5 : (DATA == 1 ? u - 1.5 : u - 1.0);
  }
  __syncthreads();
  unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  const long long c0 = clock64(), w0 = wall_clock64();
  if (LOADS == 0) {
    if (MODE == 0) {
      asm volatile(LOOP(MFV) : : "s"(iters), "v"(addr) : "s20", "scc", "memory", FRAGS, CL("v"));
    } else {
      asm volatile(LOOP(MFA) : : "s"(iters), "v"(addr) : "s20", "scc", "memory", FRAGS, CL("a"));
    }
  } else {
    // per-thread byte offset into src; advances 32 KB per workgroup-iteration, wraps by mask
    unsigned off = (blockIdx.x * 997u * 32768u + (threadIdx.x >> 6) * 8192u + (threadIdx.x & 63) * 16u +
                    4096u) & mask;  // src points 32 KB into its allocation
    asm volatile(
        "s_mov_b32 s20, %0\n"
        "1:\n" READS
        "s_waitcnt vmcnt(0)\n"
        "global_load_dwordx4 v[192:195], %2, %3 offset:-4096\n global_load_dwordx4 v[196:199], %2, %3 offset:-3072\n"
        "global_load_dwordx4 v[200:203], %2, %3 offset:-2048\n global_load_dwordx4 v[204:207], %2, %3 offset:-1024\n"
        "global_load_dwordx4 v[208:211], %2, %3\n global_load_dwordx4 v[212:215], %2, %3 offset:1024\n"
        "global_load_dwordx4 v[216:219], %2, %3 offset:2048\n global_load_dwordx4 v[220:223], %2, %3 offset:3072\n"
        "s_waitcnt lgkmcnt(0)\n" GROUPX(MFV) GROUPY(MFV)
        "v_xor_b32 %1, 0x4000, %1\n"
        "v_add_u32 %2, 0x8000, %2\n"
        "v_and_b32 %2, %4, %2\n"
        "s_sub_u32 s20, s20, 1\n"
        "s_cmp_lg_u32 s20, 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt vmcnt(0)\n"
        "s_nop 15\n"
        :
        : "s"(iters), "v"(addr), "v"(off), "s"(src), "v"(mask)
        : "s20", "scc", "memory", FRAGS, CL("v"), "v192", "v193", "v194", "v195", "v196", "v197",
          "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208",
          "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219",
          "v220", "v221", "v222", "v223");
  }
  if (threadIdx.x == 0) {
    dbg[2 * blockIdx.x] = (double)(clock64() - c0);
    dbg[2 * blockIdx.x + 1] = (double)(wall_clock64() - w0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (double)addr;
}

template <int MODE, int DATA, int LOADS>
void run(int cus, double* out, double* dbg, const double* src) {
  const unsigned mask = LOADS == 1 ? (2u << 20) - 1 : (1u << 30) - 1;
  const int iters = 20000;  // x 32 MFMAs
  dim3 grid(2 * cus), block(256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<MODE, DATA, LOADS>), grid, block, 0, 0, out, 100, dbg, src + 4096, mask);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<MODE, DATA, LOADS>), grid, block, 0, 0, out, iters, dbg, src + 4096, mask);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> hdbg(2 * grid.x);
    (void)hipMemcpy(hdbg.data(), dbg, sizeof(double) * 2 * grid.x, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (unsigned i = 0; i < grid.x; ++i) { c += hdbg[2 * i]; w += hdbg[2 * i + 1]; }
    const double flops = (double)grid.x * 4 * iters * 32 * 2048.0;
    printf("acc in %s, %s LDS data, %s: %.3f ms  %.2f TFLOP/s  cycles/MFMA %.1f  shader clock %.0f MHz  %s\n",
           MODE ? "AccVGPRs " : "arch VGPRs",
           DATA == 0 ? "constant       " : (DATA == 1 ? "random +-0.5   " : "random [0,1)   "),
           LOADS == 0 ? "no global loads" : (LOADS == 1 ? "loads from L2  " : "loads from HBM "), ms, flops / ms / 1e9,
           c / grid.x / (2.0 * iters * 32), c / w * 100.0, hipGetErrorString(hipGetLastError()));
  }
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  double *out, *dbg;
  (void)hipMalloc(&out, 8 * 256 * 4096);
  (void)hipMalloc(&dbg, 8 * 2 * 4096);
  double* src;
  (void)hipMalloc(&src, (1u << 30) + 65536);
  {  // random doubles in [0, 1)
    std::vector<double> h((1u << 27) + 8192);
    unsigned long long x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0; }
    (void)hipMemcpy(src, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  }
  run<0, 0, 0>(p.multiProcessorCount, out, dbg, src);
  run<0, 1, 0>(p.multiProcessorCount, out, dbg, src);
  run<0, 2, 0>(p.multiProcessorCount, out, dbg, src);
  run<0, 2, 1>(p.multiProcessorCount, out, dbg, src);
  run<0, 2, 2>(p.multiProcessorCount, out, dbg, src);
  run<0, 0, 2>(p.multiProcessorCount, out, dbg, src);
  return 0;
}
