# Round-end evidence run (on the GPU box: gpurun -- bash tests/probes/round_end_evidence.sh): GPU tests, default bench, rocprofv3
# kernel stats of the same bench command, FETCH/WRITE PMC passes -> gpurun_out/final/
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-600 $O/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --no-extras --no-concurrent --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB > $O/kernel_stats.txt
head -12 $O/kernel_stats.txt
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2> $O/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2> $O/pmc_write.err
python tools/make_pmc_traffic.py $O/pmc_fetch $O/pmc_write "round 2 final" > $O/pmc_traffic_diffuse.json
cat $O/pmc_traffic_diffuse.json
python tools/pmc_summary.py $(dirname $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)) $(dirname $(find $O/pmc_write -name '*counter_collection.csv' | head -1)) > $O/pmc.txt
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
du -sh $O
