cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
timeout 1500 python -m pytest tests/test_gpu_batch_grouped.py -x -q > gpurun_out/g2/pytest.log 2>&1; echo "exit $?" >> gpurun_out/g2/pytest.log
tail -25 gpurun_out/g2/pytest.log
timeout 600 python tools/batch_group_probe.py 8 16 > gpurun_out/g2/probe2.log 2>&1
cat gpurun_out/g2/probe2.log
SC_GROUP_FRONT_BY_MEMBER=1 timeout 600 python tools/group_only.py 16 | tail -1
