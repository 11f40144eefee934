"""Where a workgroup slot's time goes in the Diffuse GEMM (n = 8192), from the per-tile
records of SC_GEMM_CLOCK=<file> (workgroup item, cycles, ticks entry..K-loop end, start tick,
prologue ticks, then ticks since the end of the K loop: statistics done, stores issued,
stores drained; 100 MHz ticks):
   python tests/probes/gemm_gap_probe.py [n]
"""
import os
import subprocess
import sys
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
n = sys.argv[1] if len(sys.argv) > 1 else "8192"
path = "/tmp/gemm_tiles.txt"
env = dict(os.environ, SC_GEMM_CLOCK=path)
subprocess.run([sys.executable, os.path.join(here, "diffuse_only.py"), n, "3", "sparse"], env=env,
               check=True)
d = np.loadtxt(path)
b = d[:, 0].astype(int)
ticks, start, pro, stats, issued, drained = (d[:, i] / 100.0 for i in (2, 3, 4, 5, 6, 7))
start -= start.min()
kloop = ticks - pro
end = start + ticks + drained
print("tiles %d  launch span %.0f us (first start .. last drained)" % (len(b), end.max()))
for name, v in (("prologue", pro), ("K loop", kloop), ("statistics", stats),
                ("stores issued", issued - stats), ("stores drained", drained - issued)):
  print("  %-14s mean %8.2f us  min %8.2f  max %8.2f" % (name, v.mean(), v.min(), v.max()))
busy = (ticks + drained).sum()
print("  sum(prologue + K loop + epilogue) = %.0f us = %.1f slots x span" % (busy, busy / end.max()))
print("  sum(K loop) / (512 x span) = %.3f" % (kloop.sum() / (512 * end.max())))
# per XCD: chains of tiles on a slot cannot be told apart, but the gaps can: sort the starts
# and ends of an XCD's tiles; the k-th start after the first 64 follows the k-th end
xcd = b & 7
gaps = []
for x in range(8):
  m = xcd == x
  s = np.sort(start[m])
  e = np.sort(end[m])
  k = len(s) - 64
  if k > 0:
    gaps.append(s[64:] - e[:k])
g = np.concatenate(gaps)
print("  end of a tile's stores -> start of the next tile on the slot (order statistics): "
      "mean %.2f us, median %.2f, p90 %.2f" % (g.mean(), np.median(g), np.percentile(g, 90)))
# the tail: when do the slots run out of whole tiles (the 512 latest ends), against the span
# of the launch as the host sees it (split-K units of the 32 leftover tiles fill the rest)
last = np.sort(end)[-512:]
print("  last whole tile of a slot ends at: min %.0f  mean %.0f  max %.0f us" %
      (last.min(), last.mean(), last.max()))
for x in range(8):
  m = xcd == x
  print("   XCD %d: tiles %d, K loop mean %.0f us, last end %.0f" %
        (x, m.sum(), kloop[m].mean(), end[m].max()))
gen = np.argsort(start)
for g in range(4):
  sl = gen[512 * g:512 * (g + 1)]
  print("   generation %d: start %.0f..%.0f, K loop mean %.0f us (min %.0f max %.0f)" %
        (g, start[sl].min(), start[sl].max(), kloop[sl].mean(), kloop[sl].min(), kloop[sl].max()))
