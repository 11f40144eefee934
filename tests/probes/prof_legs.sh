# rocprofv3 kernel stats of bench.py's side legs (the Turn-to-Diarize sweep, hard8192):
#   gpurun -- bash tests/probes/prof_legs.sh [leg ...]   -> gpurun_out/legs/<leg>_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/legs; mkdir -p $O
for leg in ${@:-ttd_sweep_trace hard8192_trace}; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$leg -o run -- python $R/tests/probes/$leg.py > $O/$leg.log 2>&1
  DB=$(ls $O/prof_$leg/*/*.db $O/prof_$leg/*.db 2>/dev/null | head -1)
  python $R/tools/rocprof_summary.py $DB > $O/${leg}_kernel_stats.txt
  head -12 $O/${leg}_kernel_stats.txt
  rm -rf $O/prof_$leg
done
