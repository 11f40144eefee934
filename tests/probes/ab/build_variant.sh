#!/bin/bash
# A/B of GEMM kernel variants inside ONE gpurun call (boxes differ by +-1 % in clock, so
# cross-call comparisons cannot resolve a 1 % effect):
#   bash tests/probes/ab/build_variant.sh <name> <patched gemm_f64.hip>   (here, no GPU)
# builds tests/probes/ab/lib_<name>.so from the product objects with the given GEMM source;
# tests/probes/ab/run_ab.sh then benches every lib_*.so through SPECTRALCLUSTER_AMD_LIB.
set -e
NAME=$1; SRC=$2
ROOT=$(cd $(dirname $0)/../../.. && pwd)
C=$ROOT/spectralcluster_amd/csrc
OUT=$ROOT/tests/probes/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value \
  -Wno-unused-result -I$C -I$ROOT/include -c $SRC -o $OUT/gemm_$NAME.o 2> $OUT/build_$NAME.log || { tail -20 $OUT/build_$NAME.log; exit 1; }
OBJS=$(ls $C/*.o | grep -v gemm_f64.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/gemm_$NAME.o -ldl -pthread -o $OUT/lib_$NAME.so
rm -f $OUT/gemm_$NAME.o $OUT/build_$NAME.log
ls -la $OUT/lib_$NAME.so
