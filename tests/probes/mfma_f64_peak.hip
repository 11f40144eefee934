// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on this chip (no memory
// traffic, whole loop in one asm block so hipcc cannot add AGPR<->VGPR copies).
// This is the real ceiling the fp64 GEMM is measured against (DVFS included).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

#define M(i) "v_mfma_f64_16x16x4_f64 a[" #i ":" #i "+7], %1, %2, a[" #i ":" #i "+7]\n"

__global__ __launch_bounds__(256) void k_peak16(double* out, int iters, double a0, double b0) {
  double a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5;
  asm volatile(
      "s_mov_b32 s20, %0\n"
      "1:\n" M(0) M(8) M(16) M(24) M(32) M(40) M(48) M(56) M(64) M(72) M(80) M(88) M(96)
          M(104) M(112) M(120)
      "s_sub_u32 s20, s20, 1\n"
      "s_cmp_lg_u32 s20, 0\n"
      "s_cbranch_scc1 1b\n"
      "s_nop 15\n"
      :
      : "s"(iters), "v"(a), "v"(b)
      : "s20", "scc", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10",
        "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22",
        "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34",
        "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46",
        "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58",
        "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70",
        "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82",
        "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94",
        "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105",
        "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115",
        "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125",
        "a126", "a127");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}

void run(int blocks_per_cu, int cus, double* out) {
  const int iters = 20000;
  dim3 grid(blocks_per_cu * cus), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_peak16, grid, block, 0, 0, out, 100, 1.0, 2.0);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak16, grid, block, 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid.x * 4 /*waves*/ * iters * 16 * 2048.0;
    printf("16 acc, blocks/CU=%d: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n",
           blocks_per_cu, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * iters * 16));
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  double* out;
  hipMalloc(&out, 8 * 256 * 4096);
  run(1, p.multiProcessorCount, out);
  run(2, p.multiProcessorCount, out);
  return 0;
}
