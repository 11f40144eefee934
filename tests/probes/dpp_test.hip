// Semantics of the DPP helpers of spectralcluster_amd/csrc/dpp.h on the device: whole-wave
// neighbour moves, the wave maximum and the half-wave sums.
//   hipcc --offload-arch=gfx950 -O3 -I spectralcluster_amd/csrc -o tools/bin/dpp_test tests/probes/dpp_test.hip
#include "dpp.h"

#include <cstdio>

using namespace sc;

__global__ void k(double* out, int* iout, const double* in) {
  const double v = in[threadIdx.x];
  out[threadIdx.x] = lane_from_below(v);
  out[64 + threadIdx.x] = lane_from_above(v);
  out[128 + threadIdx.x] = wave_max_to_last(v);
  out[192 + threadIdx.x] = half_sum_to_last(v);
  iout[threadIdx.x] = half_sum_to_last((int)v);
}

int main() {
  double *in, *out;
  int* iout;
  (void)hipMalloc(&in, 64 * 8);
  (void)hipMalloc(&out, 256 * 8);
  (void)hipMalloc(&iout, 64 * 4);
  double h[64], o[256];
  int io[64];
  for (int i = 0; i < 64; ++i) h[i] = (i * 37) % 64 + 0.5;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, iout, in);
  (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  (void)hipMemcpy(io, iout, sizeof(io), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 1; i < 64; ++i) bad += o[i] != h[i - 1];
  for (int i = 0; i < 63; ++i) bad += o[64 + i] != h[i + 1];
  bad += o[0] != h[0];
  bad += o[127] != h[63];
  double mx = 0, s0 = 0, s1 = 0;
  int i0 = 0, i1 = 0;
  for (int i = 0; i < 64; ++i) mx = mx > h[i] ? mx : h[i];
  for (int i = 0; i < 32; ++i) { s0 += h[i]; i0 += (int)h[i]; }
  for (int i = 32; i < 64; ++i) { s1 += h[i]; i1 += (int)h[i]; }
  bad += o[128 + 63] != mx;
  bad += o[192 + 31] != s0;
  bad += o[192 + 63] != s1;
  bad += io[31] != i0;
  bad += io[63] != i1;
  printf("dpp helpers: %d errors (max %g, half sums %g %g, int %d %d)\n", bad, o[128 + 63], o[192 + 31],
         o[192 + 63], io[31], io[63]);
  return bad != 0;
}
