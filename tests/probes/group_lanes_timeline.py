"""Timeline summary of the last grouped-batch pass in a rocprofv3 kernel-trace database:
   python tests/probes/group_lanes_timeline.py <results.db>
per stream busy time, time with 0 / 1 / >= 2 GEMM launches in flight, idle time."""
import collections
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# the last pass: after the last gap > 20 ms? simpler: the last 'fraction' of the trace
lo = rows[0][1] + float(sys.argv[2] if len(sys.argv) > 2 else 0.75) * (rows[-1][2] - rows[0][1])
rows = [r for r in rows if r[1] >= lo]
t0, t1 = rows[0][1], max(r[2] for r in rows)
span = (t1 - t0) / 1e3
by = collections.defaultdict(lambda: [0, 0.0, 0.0, set()])
for n, s, e, st, q in rows:
  b = by[st]
  b[0] += 1
  b[1] += (e - s) / 1e3
  if "gemm_nt" in n:
    b[2] += (e - s) / 1e3
  b[3].add(q)
print("window %.1f ms, %d launches" % (span / 1e3, len(rows)))
for st, (c, busy, gemm, qs) in sorted(by.items()):
  print("  stream %3d (queue %s): %4d kernels, busy %5.1f ms (%2.0f %%), gemm %5.1f ms" %
        (st, ",".join(str(q) for q in sorted(qs)), c, busy / 1e3, 100 * busy / span, gemm / 1e3))
ev = []
for n, s, e, st, q in rows:
  ev.append((s, 1, "gemm_nt" in n))
  ev.append((e, -1, "gemm_nt" in n))
ev.sort()
run = g = 0
last = t0
acc = collections.Counter()
for t, d, is_g in ev:
  key = ("idle" if run == 0 else "no gemm" if g == 0 else "1 gemm" if g == 1 else ">=2 gemm")
  acc[key] += t - last
  last = t
  run += d
  if is_g:
    g += d
print("  " + "   ".join("%s %.1f ms (%.0f %%)" % (k, v / 1e6, 100 * v / 1e3 / span)
                        for k, v in sorted(acc.items())))
