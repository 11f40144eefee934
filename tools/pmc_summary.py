#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <dir>..."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
  for path in glob.glob(d + "/*counter_collection.csv"):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(dict)
    for r in rows:
      k = r["Kernel_Name"].split("(")[0].replace("void ", "")
      agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
      disp[k].add(r["Dispatch_Id"])
      dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("# %s" % path)
    for k in agg:
      nd = len(disp[k])
      print("%-45s dispatches=%d avg_us=%.1f  " % (k[:45], nd, sum(dur[k].values()) / nd) +
            "  ".join("%s=%.6g" % (c, v / nd) for c, v in sorted(agg[k].items())))
