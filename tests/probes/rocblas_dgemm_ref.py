"""Reference point for the Diffuse GEMM: what the vendor library (torch.mm -> rocBLAS /
hipBLASLt) reaches for a full fp64 8192^3 GEMM on this device.  Not on the product path."""
import time
import torch
n = 8192
a = torch.rand(n, n, dtype=torch.float64, device="cuda")
b = torch.rand(n, n, dtype=torch.float64, device="cuda")
for _ in range(3):
  c = a @ b.T
torch.cuda.synchronize()
t = time.time()
reps = 10
for _ in range(reps):
  c = a @ b.T
torch.cuda.synchronize()
dt = (time.time() - t) / reps
print("torch fp64 NT gemm n=%d: %.2f ms, %.1f TFLOP/s" % (n, dt * 1e3, 2 * n**3 / dt / 1e12))
