#!/bin/bash
# on the GPU box: HBM fetch of the Diffuse GEMM per launch, per variant library
# (rocprofv3 --pmc FETCH_SIZE, kernel trace only; x2 = the gfx950 correction of the guide)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in spectralcluster_amd/csrc/libspectralcluster_amd.so tests/probes/ab/lib_*.so; do
  O=$GRAFT_REPO_ROOT/gpurun_out/abpmc; rm -rf $O; mkdir -p $O
  (cd /tmp && SPECTRALCLUSTER_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-extras --no-concurrent --no-cpu-baseline > /dev/null 2>&1)
  python - "$lib" $(find $O -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_gemm_nt" in r["Kernel_Name"] and "Li0ELb1" in r["Kernel_Name"] or "k_gemm_nt<0, true>" in r["Kernel_Name"]]
v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "FETCH_SIZE"]
print("%-40s launches %d  FETCH_SIZE mean %.3e KiB -> %.2f GB (x2)  min %.2f max %.2f" % (sys.argv[1].split("/")[-1], len(v), sum(v) / len(v), 2 * 1024 * sum(v) / len(v) / 1e9, 2 * 1024 * min(v) / 1e9, 2 * 1024 * max(v) / 1e9))
PY
done
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/abpmc
