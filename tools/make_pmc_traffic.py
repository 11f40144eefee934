#!/usr/bin/env python
"""profiles/pmc_traffic_diffuse.json from rocprofv3 --pmc passes of bench.py:
   python tools/make_pmc_traffic.py <pmc_FETCH_dir> <pmc_WRITE_dir> <label> > profiles/pmc_traffic_diffuse.json
HBM-side bytes per Diffuse launch = FETCH_SIZE [KiB] x 1024 x 2 (gfx950: the counter reports
half the bytes of 16 B/lane streaming loads, MI355X_MICROARCH.md "HBM") + WRITE_SIZE [KiB]
x 1024, averaged over the dispatches of k_gemm_nt<0, true> (+ its split-K reduce kernel)."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(d, counter):
  tot = collections.defaultdict(float)
  disp = collections.defaultdict(set)
  for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
      if r["Counter_Name"] != counter:
        continue
      k = r["Kernel_Name"].split("(")[0].replace("void ", "")
      tot[k] += float(r["Counter_Value"])
      disp[k].add(r["Dispatch_Id"])
  return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}


fetch = per_dispatch(sys.argv[1], "FETCH_SIZE")
write = per_dispatch(sys.argv[2], "WRITE_SIZE")
label = sys.argv[3] if len(sys.argv) > 3 else ""
# which kernel: "diffuse" (the fp64 product, default), "affinity" or "i8" (the digit product of the
# matrix-free Diffuse: reads n^2 * 2 B of digits, writes the fp32 upper-triangle tiles)
which = sys.argv[4] if len(sys.argv) > 4 else "diffuse"
n = 8192
if which == "affinity":
  # the cosine-affinity GEMM: reads the normalised embeddings (n d * 8 B), writes n^2 * 8 B
  main, red = "sc::k_gemm_nt<1, true>", "sc::k_gemm_reduce<1, true>"
  algorithmic = n * n * 8 + n * 256 * 8
  what = "k_gemm_nt<EpiAffinity,SYM> (+ split-K reduce)"
elif which == "i8":
  main, red = "sc::k_gemm_i8_sym<0>", "-"
  # (with the skip list: the digits of the tile rows that have a surviving tile -- all of them,
  #  every diagonal tile survives -- and the fp32 tiles that ran; tiles_run from argv[5])
  tiles_run = int(sys.argv[5]) if len(sys.argv) > 5 else (n // 128) * (n // 128 + 1) // 2
  algorithmic = n * n * 2 + tiles_run * 128 * 128 * 4
  what = "k_gemm_i8_sym"
else:
  main, red = "sc::k_gemm_nt<0, true>", "sc::k_gemm_reduce<0, true>"
  algorithmic = 2 * n * n * 8
  what = "k_gemm_nt<EpiNone,SYM> (+ split-K reduce)"
f = fetch[main][0] + (fetch[red][0] if red in fetch else 0.0)
w = write[main][0] + (write[red][0] if red in write else 0.0)
print(json.dumps({
    "kernel": what,
    "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024),
    "source": "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 3 "
              "--no-extras`, per-dispatch average over %d launches of %s at n=8192: FETCH_SIZE "
              "%.4g KiB x 2 (gfx950 16 B/lane correction) + WRITE_SIZE %.4g KiB"
              % (label, fetch[main][1], what, f, w),
    "algorithmic_bytes_per_launch": algorithmic,
    # what it was measured on (bench.py compares the hash with the current source)
    "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"],
                             capture_output=True, text=True).stdout.strip() or None,
    "gemm_f64_sha16": hashlib.sha256(open(os.path.join(
        ROOT, "spectralcluster_amd", "csrc", "gemm_f64.hip"), "rb").read()).hexdigest()[:16],
    "diffuse_free_sha16": hashlib.sha256(open(os.path.join(
        ROOT, "spectralcluster_amd", "csrc", "diffuse_free.hip"), "rb").read()).hexdigest()[:16],
    "per_kernel_bytes": {k: int(fetch[k][0] * 2048 + (write[k][0] if k in write else 0.0) * 1024)
                         for k in sorted(fetch)},
}, indent=1))
