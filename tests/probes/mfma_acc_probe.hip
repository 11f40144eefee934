// Microbenchmark (not on the product path): does v_mfma_f64_16x16x4_f64 run slower when the
// accumulators live in architectural VGPRs (what hipcc picks for csrc/gemm_f64.hip under
// __launch_bounds__(256, 2)) than in AccVGPRs (what the vendor's Tensile kernels use)?
// 16 accumulators, 4 A and 4 B operand pairs in distinct registers like a 64 x 64 wave tile.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_acc_probe.hip -o tools/bin/mfma_acc_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define MV(i, a, b) "v_mfma_f64_16x16x4_f64 v[" #i ":" #i "+7], " a ", " b ", v[" #i ":" #i "+7]\n"
#define MA(i, a, b) "v_mfma_f64_16x16x4_f64 a[" #i ":" #i "+7], " a ", " b ", a[" #i ":" #i "+7]\n"
#define BODY(M)                                                                          \
  M(0, "%1", "%5") M(8, "%1", "%6") M(16, "%1", "%7") M(24, "%1", "%8")                  \
  M(32, "%2", "%5") M(40, "%2", "%6") M(48, "%2", "%7") M(56, "%2", "%8")                \
  M(64, "%3", "%5") M(72, "%3", "%6") M(80, "%3", "%7") M(88, "%3", "%8")                \
  M(96, "%4", "%5") M(104, "%4", "%6") M(112, "%4", "%7") M(120, "%4", "%8")
#define LOOP(M)                                                                          \
  "s_mov_b32 s20, %0\n"                                                                   \
  "1:\n" BODY(M) "s_sub_u32 s20, s20, 1\n"                                               \
  "s_cmp_lg_u32 s20, 0\n"                                                                 \
  "s_cbranch_scc1 1b\n"                                                                   \
  "s_nop 15\n"
#define R8(p, i) p #i
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

template <int ACC>
__global__ __launch_bounds__(256, 2) void k_peak(double* out, int iters, double a0, double b0) {
  const double t = threadIdx.x;
  double a1 = a0 + t, a2 = a0 - t, a3 = a0 * 0.5 + t, a4 = a0 + 2 * t;
  double b1 = b0 + t, b2 = b0 - t, b3 = b0 * 0.5 + t, b4 = b0 + 2 * t;
  if (ACC == 0) {
    asm volatile(LOOP(MV)
                 :
                 : "s"(iters), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b1), "v"(b2), "v"(b3),
                   "v"(b4)
                 : "s20", "scc", CL("v"));
  } else {
    asm volatile(LOOP(MA)
                 :
                 : "s"(iters), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b1), "v"(b2), "v"(b3),
                   "v"(b4)
                 : "s20", "scc", CL("a"));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a1 + b1;
}

template <int ACC>
void run(int blocks_per_cu, int cus, double* out) {
  const int iters = 20000;
  dim3 grid(blocks_per_cu * cus), block(256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_peak<ACC>, grid, block, 0, 0, out, 100, 1.0, 2.0);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak<ACC>, grid, block, 0, 0, out, iters, 1.0, 2.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid.x * 4 * iters * 16 * 2048.0;
    printf("acc in %s, workgroups/CU=%d: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA at 2.4 GHz)\n",
           ACC ? "AccVGPRs " : "arch VGPRs", blocks_per_cu, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * iters * 16));
  }
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  double* out;
  (void)hipMalloc(&out, 8 * 256 * 4096);
  run<0>(1, p.multiProcessorCount, out);
  run<1>(1, p.multiProcessorCount, out);
  run<0>(2, p.multiProcessorCount, out);
  run<1>(2, p.multiProcessorCount, out);
  return 0;
}
