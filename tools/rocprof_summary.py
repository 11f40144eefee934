#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the text summary we
commit under profiles/:  python tools/rocprof_summary.py <results.db> > profiles/x.txt"""
import sqlite3
import sys


def short_name(name):
  """kernel name without its argument list; a kernel in an unnamed namespace keeps its name
  ("(anonymous namespace)::k(...)" used to come out empty: the 5.9 % row without a name in
  profiles/r06k_grouped_kernel_stats.txt)"""
  name = name.replace("void ", "").replace("(anonymous namespace)::", "")
  return name.split("(")[0]


def main(path):
  cur = sqlite3.connect(path).cursor()
  rows = list(cur.execute(
      "select name, total_calls, total_duration, average, percentage from top_kernels"))
  print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
  print("# durations in microseconds")
  print("%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
  for name, calls, total, avg, pct in rows:
    short = short_name(name)
    print("%-70s %8d %14.1f %12.2f %7.2f" % (short[:70], calls, total, avg, pct))
  extra = list(cur.execute(
      "select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, "
      "workgroup_x from kernels group by name"))
  print("\n# per-kernel resources (vgpr, agpr, sgpr, lds bytes, grid_x, wg_x)")
  for r in extra:
    print("%-70s %s" % (short_name(r[0])[:70], " ".join(str(v) for v in r[1:])))


if __name__ == "__main__":
  main(sys.argv[1])
