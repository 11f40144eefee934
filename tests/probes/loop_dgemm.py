import sys, time, torch
n = 8192
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
a = torch.rand(n, n, dtype=torch.float64, device="cuda"); b = torch.rand(n, n, dtype=torch.float64, device="cuda")
if mode == 1:
  a.fill_(1.0); b.fill_(0.5)
torch.cuda.synchronize(); t = time.time(); reps = 0
while time.time() - t < 3.0:
  for _ in range(10): c = a @ b.T
  torch.cuda.synchronize(); reps += 10
dt = (time.time() - t) / reps
print("torch dgemm mode %d: %.2f ms %.1f TF/s" % (mode, dt * 1e3, 2 * n**3 / dt / 1e12))
