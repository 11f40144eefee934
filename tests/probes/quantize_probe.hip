// What the quantiser of the matrix-free Diffuse (k_free_quantize) spends its time on: the
// shipped kernel against variants without the same-address atomicMax, without the digit
// conversion, without the copy-out of the row image -- and a plain copy of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spectralcluster_amd/csrc \
//     -o tools/bin/quantize_probe tests/probes/quantize_probe.hip
#include "diffuse_free.hip"

#include <cstdio>
#include <vector>

using namespace sc;

template <int PROBE>
static float run(const double* A, int n, int ld, signed char* Q, double* scal, double* y1, double* R) {
  const int Kp = free_k_padded(n);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_free_quantize<PROBE>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_free_quantize<PROBE>, dim3(free_rows_padded(n)), dim3(256), (size_t)2 * Kp,
                       0, A, n, ld, Q, (size_t)2 * Kp, Kp, scal, y1, R,
                       reinterpret_cast<unsigned long long*>(scal) + 2);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t count) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192;
  const int ld = n + 16;
  std::vector<double> h((size_t)n * ld);
  unsigned x = 1;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) * (1.0 / 16777216.0); }
  double *A, *scal, *y1, *R, *B;
  signed char* Q;
  hipMalloc(&A, h.size() * 8);
  hipMalloc(&B, h.size() * 8);
  hipMalloc(&scal, 64);
  hipMalloc(&y1, n * 8);
  hipMalloc(&R, n * 8);
  hipMalloc(&Q, free_q_bytes(n));
  hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  const double one[4] = {1.0, 0, 0, 0};
  hipMemcpy(scal, one, 32, hipMemcpyHostToDevice);
  const double gb = ((double)n * n * 8 + free_q_bytes(n)) / 1e9;
  const float t0 = run<0>(A, n, ld, Q, scal, y1, R);
  const float t1 = run<1>(A, n, ld, Q, scal, y1, R);
  const float t2 = run<2>(A, n, ld, Q, scal, y1, R);
  const float t3 = run<3>(A, n, ld, Q, scal, y1, R);
  printf("n=%d: quantiser %.1f us (%.2f TB/s of %.2f GB); no atomicMax %.1f; no digits %.1f; no copy-out %.1f\n",
         n, t0 * 1e3, gb / t0 * 1e-3 * 1e3 / 1e3, gb, t1 * 1e3, t2 * 1e3, t3 * 1e3);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {2048, 8192, 32768}) {
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const double2*>(A),
                         reinterpret_cast<double2*>(B), (size_t)n * ld / 2);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
    printf("  plain copy of A (read + write %.2f GB), grid %d: %.1f us = %.2f TB/s\n",
           2.0 * n * ld * 8 / 1e9, grid, best * 1e3, 2.0 * n * ld * 8 / 1e9 / best);
  }
  return 0;
}
