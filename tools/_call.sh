cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --no-extras --no-concurrent --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB > $O/kernel_stats.txt
head -14 $O/kernel_stats.txt
rm -rf $O/prof
