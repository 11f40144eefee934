cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
timeout 1500 python -m pytest tests/test_gpu_batch_grouped.py tests/test_gpu_edge_cases.py tests/test_gpu_stages.py tests/test_gpu_baseline_configs.py tests/test_gpu_predict.py -x -q > gpurun_out/g2/pytest.log 2>&1; echo "exit $?" >> gpurun_out/g2/pytest.log
tail -15 gpurun_out/g2/pytest.log
SC_GROUP_TRACE=1 timeout 600 python tools/group_only.py 16 > gpurun_out/g2/probe.log 2> gpurun_out/g2/trace.log
cat gpurun_out/g2/probe.log
grep "group eigen" gpurun_out/g2/trace.log | tail -32 | awk 'NR%4==1'
grep "group of" gpurun_out/g2/trace.log | tail -32 | awk 'NR%4==1'
timeout 600 python tools/batch_group_probe.py 8 16 > gpurun_out/g2/probe2.log 2>&1
cat gpurun_out/g2/probe2.log
