#!/bin/bash
# on the GPU box: bash tests/probes/ab/run_ab.sh [reps] -- the default bench (n = 8192, no
# extras) once per variant library and repetition, interleaved
cd $GRAFT_REPO_ROOT
REPS=${1:-2}
B="python bench.py --steps 30 --warmup 5 --no-extras --no-concurrent --no-cpu-baseline"
for rep in $(seq $REPS); do
  for lib in spectralcluster_amd/csrc/libspectralcluster_amd.so tests/probes/ab/lib_*.so; do
    SPECTRALCLUSTER_AMD_LIB=$PWD/$lib $B 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=b['stage_ms']
print('%-44s call %.3f ms  frac %.4f  diffuse %.3f  affinity %.3f' % ('$lib'.split('/')[-1], b['ms_per_step'], b['roofline']['frac'], s['diffuse'], s['affinity']))"
  done
done
