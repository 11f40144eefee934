// What does the streaming blur (k_gaussian_blur_stream<4>) spend on WRITING the blurred matrix?
// The same launch with out = nullptr computes everything (row maxima included) and stores
// nothing -- the first pass of a "recompute the blur inside the threshold pass" scheme.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spectralcluster_amd/csrc \
//     -o tools/bin/blur_probe tests/probes/blur_probe.hip
#include "blur.hip"

#include <cstdio>
#include <vector>

using namespace sc;

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192;
  const int ld = n + 16;
  std::vector<double> h((size_t)n * ld);
  unsigned x = 1;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) * (1.0 / 16777216.0); }
  double *A, *B, *w, *diag, *rm;
  hipMalloc(&A, h.size() * 8);
  hipMalloc(&B, h.size() * 8);
  hipMalloc(&w, 64 * 8);
  hipMalloc(&diag, n * 8);
  hipMalloc(&rm, (size_t)n * blur_tile_columns(n, 4) * 8);
  hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  const double wt[9] = {1.3383e-4, 4.4319e-3, 5.3991e-2, 2.41971e-1, 3.98943e-1, 2.41971e-1, 5.3991e-2, 4.4319e-3, 1.3383e-4};
  hipMemcpy(w, wt, sizeof(wt), hipMemcpyHostToDevice);
  hipMemset(diag, 0, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
      hipEventRecord(e0, 0);
      launch_gaussian_blur_fused(0, A, mode ? nullptr : B, n, ld, 4, w, diag, rm);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
    printf("n=%d blur %s: %.1f us\n", n, mode ? "row maxima only (no store)" : "with store          ", best * 1e3);
  }
  return 0;
}
