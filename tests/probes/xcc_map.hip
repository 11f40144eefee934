// Which XCD does workgroup b land on?  (assumption used by the GEMM tile order: b % 8)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) {
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    out[blockIdx.x] = id & 0xf;
  }
}
int main() {
  const int nb = 2048;
  int* d; hipMalloc(&d, nb * 4);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d);
  int h[nb]; hipMemcpy(h, d, nb * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 48; ++i) printf("%d ", h[i]); printf("\n");
  int bad = 0; for (int i = 0; i < nb; ++i) bad += (h[i] != i % 8);
  printf("blocks not on XCD b%%8: %d of %d\n", bad, nb);
  return 0;
}
