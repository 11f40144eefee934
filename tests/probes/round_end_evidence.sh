# Round-end evidence run (on the GPU box: gpurun -- bash tests/probes/round_end_evidence.sh [tag]):
# GPU tests, default bench, rocprofv3 kernel stats of the same bench command, FETCH/WRITE PMC
# passes (separate runs, kernel-trace only) -> gpurun_out/<tag>/
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-final}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-concurrent --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err)
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB > $O/kernel_stats.txt
head -14 $O/kernel_stats.txt
(cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extras --no-concurrent --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_fetch.err)
(cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_write -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extras --no-concurrent --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_write.err)
python tools/make_pmc_traffic.py $O/pmc_fetch $O/pmc_write "$TAG" > $O/pmc_traffic_diffuse.json
cat $O/pmc_traffic_diffuse.json
TILES=$(python -c "import json;d=json.load(open('$O/bench_under_rocprof.json'));r=d['roofline'];print(r.get('i8_product',r).get('tiles_run',2080))")
python tools/make_pmc_traffic.py $O/pmc_fetch $O/pmc_write "$TAG" i8 $TILES > $O/pmc_traffic_i8.json
python tools/make_pmc_traffic.py $O/pmc_fetch $O/pmc_write "$TAG" affinity > $O/pmc_traffic_affinity.json
cat $O/pmc_traffic_i8.json $O/pmc_traffic_affinity.json
python tools/pmc_summary.py $(dirname $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)) $(dirname $(find $O/pmc_write -name '*counter_collection.csv' | head -1)) > $O/pmc.txt
# the grouped batch alone under the kernel trace (timeline of config 5)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof5 -o run -- python $GRAFT_REPO_ROOT/tests/probes/group_only.py 16 > $GRAFT_REPO_ROOT/$O/group_only.log 2>&1)
DB=$(ls $O/prof5/*/*.db $O/prof5/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB > $O/grouped_kernel_stats.txt
head -14 $O/grouped_kernel_stats.txt
timeout 300 python tests/probes/gen_dense_time.py > $O/gen_dense_time.txt 2>&1
cat $O/gen_dense_time.txt
rm -rf $O/prof $O/prof5 $O/pmc_fetch $O/pmc_write
du -sh $O
