// Microbenchmark (not on the product path): v_mfma_f64_16x16x4_f64 stream (16 accumulators in
// arch VGPRs, two workgroups per CU = two waves per SIMD, like csrc/gemm_f64.hip) with R
// independent ds_read_b128 per 64 MFMAs interleaved.  The GEMM issues 16 per 64 MFMAs.
// Question: do LDS returns into the VGPR file slow the MFMA stream down?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_probe.hip -o tools/bin/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define MV(i, a, b) "v_mfma_f64_16x16x4_f64 v[" #i ":" #i "+7], " a ", " b ", v[" #i ":" #i "+7]\n"
#define RD(r, off) "ds_read_b128 v[" #r ":" #r "+3], %9 offset:" #off "\n"
#define NONE
// 16 MFMAs with up to 8 reads spread between them
#define B16(r0, r1, r2, r3, r4, r5, r6, r7)                                                \
  MV(0, "%1", "%5") r0 MV(8, "%1", "%6") MV(16, "%1", "%7") r1 MV(24, "%1", "%8")          \
  MV(32, "%2", "%5") r2 MV(40, "%2", "%6") MV(48, "%2", "%7") r3 MV(56, "%2", "%8")        \
  MV(64, "%3", "%5") r4 MV(72, "%3", "%6") MV(80, "%3", "%7") r5 MV(88, "%3", "%8")        \
  MV(96, "%4", "%5") r6 MV(104, "%4", "%6") MV(112, "%4", "%7") r7 MV(120, "%4", "%8")
#define E NONE
#define R0 B16(E, E, E, E, E, E, E, E)
#define R2 B16(RD(160, 0), E, E, E, RD(164, 1024), E, E, E)
#define R4 B16(RD(160, 0), E, RD(164, 1024), E, RD(168, 2048), E, RD(172, 3072), E)
#define R8                                                                                 \
  B16(RD(160, 0), RD(164, 1024), RD(168, 2048), RD(172, 3072), RD(176, 4096), RD(180, 5120), \
      RD(184, 6144), RD(188, 7168))
#define CLOB                                                                               \
  "s20", "scc", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11",   \
  "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", \
  "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", \
  "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", \
  "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", \
  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", \
  "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", \
  "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",      \
  "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",    \
  "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",    \
  "v124", "v125", "v126", "v127", "v160", "v161", "v162", "v163", "v164", "v165", "v166",    \
  "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177",    \
  "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188",    \
  "v189", "v190", "v191", "memory"

// READS per 64 MFMAs: 0, 8, 16, 32
template <int READS, bool RANDOM>
__global__ __launch_bounds__(256, 2) void k_probe(double* out, int iters, double a0, double b0) {
  __shared__ __attribute__((aligned(16))) double lds[4096];
  const double t = threadIdx.x;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = t + i;
  __syncthreads();
  double a1 = a0 + t, a2 = a0 - t, a3 = a0 * 0.5 + t, a4 = a0 + 2 * t;
  double b1 = b0 + t, b2 = b0 - t, b3 = b0 * 0.5 + t, b4 = b0 + 2 * t;
  if (RANDOM) {  // full random mantissas in [1, 2): data-dependent power
    unsigned long long h = 0x9E3779B97F4A7C15ull * (blockIdx.x * 256 + threadIdx.x + 1);
    auto next = [&]() {
      h ^= h << 13; h ^= h >> 7; h ^= h << 17;
      return __longlong_as_double((long long)((h >> 12) | 0x3FF0000000000000ull));
    };
    a1 = next(); a2 = next(); a3 = next(); a4 = next();
    b1 = next(); b2 = next(); b3 = next(); b4 = next();
  }
  const unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
#define ASM(BODY)                                                                          \
  asm volatile("s_mov_b32 s20, %0\n"                                                        \
               "1:\n" BODY "s_waitcnt lgkmcnt(0)\n"                                         \
               "s_sub_u32 s20, s20, 1\n"                                                    \
               "s_cmp_lg_u32 s20, 0\n"                                                      \
               "s_cbranch_scc1 1b\n"                                                        \
               "s_nop 15\n"                                                                 \
               :                                                                           \
               : "s"(iters), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b1), "v"(b2), "v"(b3), \
                 "v"(b4), "v"(addr)                                                        \
               : CLOB)
  if (READS == 0) ASM(R0 R0 R0 R0);
  if (READS == 8) ASM(R2 R2 R2 R2);
  if (READS == 16) ASM(R4 R4 R4 R4);
  if (READS == 32) ASM(R8 R8 R8 R8);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a1 + b1;
}

template <int READS, bool RANDOM>
void run(int cus, double* out) {
  const int iters = 5000;
  dim3 grid(2 * cus), block(256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<READS, RANDOM>), grid, block, 0, 0, out, 100, 1.0, 2.0);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<READS, RANDOM>), grid, block, 0, 0, out, iters, 1.0, 2.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid.x * 4 * iters * 64 * 2048.0;
    printf("%s operands, %2d ds_read_b128 per 64 MFMAs: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA)  %s\n", RANDOM ? "random" : "smooth", READS,
           ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (2.0 * iters * 64),
           hipGetErrorString(hipGetLastError()));
  }
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  double* out;
  (void)hipMalloc(&out, 8 * 256 * 4096);
  run<0, false>(p.multiProcessorCount, out);
  run<16, false>(p.multiProcessorCount, out);
  run<32, false>(p.multiProcessorCount, out);
  run<0, true>(p.multiProcessorCount, out);
  run<16, true>(p.multiProcessorCount, out);
  run<0, false>(p.multiProcessorCount, out);
  return 0;
}
