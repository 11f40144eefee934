cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/g24
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o run -- python tools/group_only.py 16 > $O/run.log 2> $O/prof.err
cat $O/run.log
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
python tools/kernel_timeline.py $DB 0.70 > $O/timeline_group16.txt
head -8 $O/timeline_group16.txt
rm -rf $O/prof
timeout 600 python tools/batch_group_probe.py 4 8 16 > $O/probe.log 2>&1; cat $O/probe.log
timeout 300 python tools/single_sizes.py > $O/singles.log 2>&1; cat $O/singles.log
