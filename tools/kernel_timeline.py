#!/usr/bin/env python
"""GPU occupancy of a run from a rocprofv3 --kernel-trace database (ROCm 7.2 rocpd sqlite):
   python tools/kernel_timeline.py <results.db> [skip_fraction]
Over the part of the trace after `skip_fraction` (default 0.5: the second of two identical
passes): wall span, time with at least one kernel running (union of the intervals), summed
kernel time (/ union = average overlap), per-kernel totals, idle gaps by size."""
import collections
import sqlite3
import sys


def main(path, skip):
  cur = sqlite3.connect(path).cursor()
  rows = list(cur.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
  t_lo = rows[0][1] + skip * (rows[-1][2] - rows[0][1])
  rows = [r for r in rows if r[1] >= t_lo]
  span = (max(r[2] for r in rows) - rows[0][1]) / 1e3
  busy, cur_end, gaps = 0.0, rows[0][1], []
  for _, s, e, _, _ in rows:
    if s > cur_end:
      gaps.append((s - cur_end) / 1e3)
      cur_end = s
    if e > cur_end:
      busy += (e - max(s, cur_end)) / 1e3
      cur_end = e
  total = sum((r[2] - r[1]) / 1e3 for r in rows)
  print("# %s, kernels after %.0f %% of the trace" % (path, 100 * skip))
  print("span %.1f ms   busy (>= 1 kernel) %.1f ms (%.0f %%)   summed kernel time %.1f ms "
        "(overlap x%.2f)   launches %d   queues %d   streams %d" % (
            span / 1e3, busy / 1e3, 100 * busy / span, total / 1e3, total / busy, len(rows),
            len(set(r[3] for r in rows)), len(set(r[4] for r in rows))))
  edges = [5, 20, 50, 200, 1000, 1e9]
  hist = collections.Counter()
  for g in gaps:
    for e in edges:
      if g <= e:
        hist[e] += g
        break
  print("idle time by gap size (us): " + "  ".join(
      "<=%g: %.1f ms" % (e, hist[e] / 1e3) for e in edges))
  agg = collections.defaultdict(lambda: [0, 0.0])
  for name, s, e, _, _ in rows:
    k = name.split("(")[0].replace("void ", "")
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
  print("%-60s %8s %12s %10s" % ("kernel", "calls", "total_ms", "avg_us"))
  for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-60s %8d %12.2f %10.1f" % (k[:60], c, t / 1e3, t / c))


def dump(path, frac, ms):
  """every kernel of a window of `ms` milliseconds starting at `frac` of the trace"""
  cur = sqlite3.connect(path).cursor()
  rows = list(cur.execute(
      "select name, start, end, queue_id, stream_id, grid_x, grid_y, workgroup_x, lds_size "
      "from kernels order by start"))
  t0 = rows[0][1] + frac * (rows[-1][2] - rows[0][1])
  print("# window of %.1f ms at %.0f %% of %s" % (ms, 100 * frac, path))
  print("%10s %8s %5s %6s %-34s %s" % ("start_us", "dur_us", "queue", "stream", "kernel", "grid"))
  for name, s, e, q, st, gx, gy, wx, lds in rows:
    if s < t0 or s > t0 + ms * 1e6:
      continue
    k = name.split("(")[0].replace("void ", "").replace("sc::", "")
    print("%10.1f %8.1f %5d %6d %-34s %dx%d wg%d lds%d" % (
        (s - t0) / 1e3, (e - s) / 1e3, q, st, k[:34], gx // max(wx, 1), gy, wx, lds))


def listing(path, needle):
  """every launch whose kernel name contains `needle`: start, duration, grid"""
  cur = sqlite3.connect(path).cursor()
  rows = list(cur.execute(
      "select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
  t0 = rows[0][1]
  for name, s, e, gx, gy, gz, wx in rows:
    if needle in name:
      k = name.split("(")[0].replace("void ", "").replace("sc::", "")
      print("%10.2f ms %9.1f us  %-30s %dx%dx%d" % ((s - t0) / 1e6, (e - s) / 1e3, k[:30],
                                                   gx // max(wx, 1), gy, gz))


if __name__ == "__main__":
  if len(sys.argv) > 3 and sys.argv[2] == "list":
    listing(sys.argv[1], sys.argv[3])
  elif len(sys.argv) > 3 and sys.argv[2] == "dump":
    dump(sys.argv[1], float(sys.argv[3]), float(sys.argv[4]) if len(sys.argv) > 4 else 2.0)
  else:
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
