// Is a grid-wide barrier inside ONE launch cheaper than a launch boundary on MI355X?
// The eigensolver's chain (15 links per call) and k-means (22 launches) are rows of tiny
// kernels separated by grid-wide reductions; each link costs 7-13 us.  This probe runs PHASES
// dependent phases over the same data both ways:
//   (a) one launch per phase (what the library does),
//   (b) one launch, phases separated by an atomic-counter barrier with agent-scope fences,
//   (c) the same with only the workgroups of ONE XCD doing the work (blockIdx % 8 == 0),
// where every phase reads what ALL workgroups wrote in the previous one (a 64-entry partial sum
// per workgroup), like the launch-boundary reduce of eig.hip.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/grid_barrier_probe tests/probes/grid_barrier_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kThreads = 256;
constexpr int kPart = 64;  // doubles each workgroup publishes per phase

__device__ __forceinline__ double phase_work(const double* __restrict__ data, int n, int wg, int nwg,
                                             const double* __restrict__ prev, double* __restrict__ mine) {
  // ordered sum of everybody's partials of the previous phase (what a chain link's prologue does)
  __shared__ double sm[kPart];
  const int t = threadIdx.x;
  if (t < kPart) {
    double acc = 0.0;
    for (int p = 0; p < nwg; ++p) acc += prev[(size_t)p * kPart + t];
    sm[t] = acc;
  }
  __syncthreads();
  // a little row work: 128 rows x 8 doubles per workgroup
  const int rows = 128;
  double v = 0.0;
  for (int e = t; e < rows * 8; e += kThreads) {
    const int r = wg * rows + e / 8;
    if (r < n) v += data[(size_t)r * 8 + (e & 7)] * sm[e & (kPart - 1)];
  }
  // block reduce into kPart partials
  __shared__ double red[kThreads];
  red[t] = v;
  __syncthreads();
  if (t < kPart) {
    double acc = 0.0;
    for (int q = t; q < kThreads; q += kPart) acc += red[q];
    mine[t] = acc * 1e-3 + 1.0;
  }
  __syncthreads();
  return v;
}

__global__ __launch_bounds__(kThreads) void k_phase(const double* data, int n, const double* prev, double* next) {
  phase_work(data, n, blockIdx.x, gridDim.x, prev, next + (size_t)blockIdx.x * kPart);
}

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();  // release: this workgroup's partials are visible device-wide
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();  // acquire
  }
  __syncthreads();
}

template <bool ONE_XCD>
__global__ __launch_bounds__(kThreads) void k_persistent(const double* data, int n, double* bufA, double* bufB,
                                                         int phases, unsigned* counter, int nwg) {
  int wg = blockIdx.x;
  if (ONE_XCD) {
    if (wg & 7) return;
    wg >>= 3;
  }
  double* prev = bufA;
  double* next = bufB;
  for (int ph = 0; ph < phases; ++ph) {
    phase_work(data, n, wg, nwg, prev, next + (size_t)wg * kPart);
    grid_barrier(counter, (unsigned)(nwg * (ph + 1)));
    double* t = prev; prev = next; next = t;
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192;
  const int phases = argc > 2 ? atoi(argv[2]) : 20;
  const int nwg = (n + 127) / 128;
  double *data, *bufA, *bufB;
  unsigned* counter;
  hipMalloc(&data, (size_t)n * 8 * 8);
  hipMalloc(&bufA, (size_t)nwg * kPart * 8);
  hipMalloc(&bufB, (size_t)nwg * kPart * 8);
  hipMalloc(&counter, 4);
  std::vector<double> h((size_t)n * 8, 1.0), one((size_t)nwg * kPart, 1.0);
  hipMemcpy(data, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timeit = [&](const char* what, auto fn) {
    float best = 1e9f;
    for (int r = 0; r < 8; ++r) {
      hipMemcpy(bufA, one.data(), one.size() * 8, hipMemcpyHostToDevice);
      hipMemset(counter, 0, 4);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      fn();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r > 1 && ms < best) best = ms;
    }
    std::vector<double> out(kPart);
    hipMemcpy(out.data(), (phases & 1) ? bufB : bufA, kPart * 8, hipMemcpyDeviceToHost);
    printf("  %-46s %8.1f us total, %6.2f us per phase   (check %.12g)\n", what, best * 1e3,
           best * 1e3 / phases, out[3]);
  };
  printf("n=%d, %d workgroups, %d dependent phases\n", n, nwg, phases);
  timeit("one launch per phase", [&] {
    double *p = bufA, *q = bufB;
    for (int ph = 0; ph < phases; ++ph) {
      hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(kThreads), 0, 0, data, n, p, q);
      double* t = p; p = q; q = t;
    }
  });
  timeit("one launch, atomic barrier + agent fences", [&] {
    hipLaunchKernelGGL(k_persistent<false>, dim3(nwg), dim3(kThreads), 0, 0, data, n, bufA, bufB, phases,
                       counter, nwg);
  });
  timeit("the same, workgroups of one XCD only", [&] {
    hipLaunchKernelGGL(k_persistent<true>, dim3(nwg * 8), dim3(kThreads), 0, 0, data, n, bufA, bufB, phases,
                       counter, nwg);
  });
  return 0;
}
