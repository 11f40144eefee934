// How fast does a pageable 16.8 MB host buffer (the embeddings of the headline call) reach the
// device?  One hipMemcpyAsync, a 2-D copy with the row pitch the library uses, the same in
// halves / quarters from several host threads on streams of their own, and a pinned buffer for
// reference.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/h2d_probe tests/probes/h2d_probe.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const size_t rows = 8192, cols = 256, bytes = rows * cols * sizeof(double);
  std::vector<double> host(rows * cols, 1.0);
  double* dev;
  hipMalloc(&dev, bytes);
  hipStream_t st[4];
  for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto best_of = [&](const char* what, auto fn) {
    double best = 1e9;
    for (int r = 0; r < 12; ++r) {
      const double t0 = now();
      fn();
      const double t = now() - t0;
      if (r > 1 && t < best) best = t;
    }
    printf("%-44s %.3f ms  %.1f GB/s\n", what, best * 1e3, bytes / best / 1e9);
  };
  best_of("pageable, one hipMemcpyAsync + sync", [&] {
    hipMemcpyAsync(dev, host.data(), bytes, hipMemcpyHostToDevice, st[0]);
    hipStreamSynchronize(st[0]);
  });
  best_of("pageable, hipMemcpy2DAsync (pitch = width)", [&] {
    hipMemcpy2DAsync(dev, cols * 8, host.data(), cols * 8, cols * 8, rows, hipMemcpyHostToDevice, st[0]);
    hipStreamSynchronize(st[0]);
  });
  for (int parts : {2, 4}) {
    char name[64];
    snprintf(name, sizeof(name), "pageable, %d threads x %d streams", parts, parts);
    best_of(name, [&] {
      std::vector<std::thread> th;
      for (int p = 0; p < parts; ++p)
        th.emplace_back([&, p] {
          const size_t chunk = bytes / parts;
          hipMemcpyAsync(reinterpret_cast<char*>(dev) + p * chunk,
                         reinterpret_cast<const char*>(host.data()) + p * chunk, chunk,
                         hipMemcpyHostToDevice, st[p]);
          hipStreamSynchronize(st[p]);
        });
      for (auto& t : th) t.join();
    });
    snprintf(name, sizeof(name), "pageable, 1 thread, %d streams", parts);
    best_of(name, [&] {
      const size_t chunk = bytes / parts;
      for (int p = 0; p < parts; ++p)
        hipMemcpyAsync(reinterpret_cast<char*>(dev) + p * chunk,
                       reinterpret_cast<const char*>(host.data()) + p * chunk, chunk,
                       hipMemcpyHostToDevice, st[p]);
      for (int p = 0; p < parts; ++p) hipStreamSynchronize(st[p]);
    });
  }
  double* pinned;
  hipHostMalloc(reinterpret_cast<void**>(&pinned), bytes);
  memcpy(pinned, host.data(), bytes);
  best_of("pinned, one hipMemcpyAsync + sync", [&] {
    hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st[0]);
    hipStreamSynchronize(st[0]);
  });
  best_of("host memcpy into the pinned buffer (1 thread)", [&] { memcpy(pinned, host.data(), bytes); });
  best_of("memcpy into pinned (4 threads) + one DMA", [&] {
    std::vector<std::thread> th;
    for (int p = 0; p < 4; ++p)
      th.emplace_back([&, p] {
        memcpy(reinterpret_cast<char*>(pinned) + p * (bytes / 4),
               reinterpret_cast<const char*>(host.data()) + p * (bytes / 4), bytes / 4);
      });
    for (auto& t : th) t.join();
    hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st[0]);
    hipStreamSynchronize(st[0]);
  });
  best_of("hipHostRegister + DMA + unregister", [&] {
    hipHostRegister(host.data(), bytes, hipHostRegisterDefault);
    hipMemcpyAsync(dev, host.data(), bytes, hipMemcpyHostToDevice, st[0]);
    hipStreamSynchronize(st[0]);
    hipHostUnregister(host.data());
  });
  return 0;
}
