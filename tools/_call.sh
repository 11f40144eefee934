cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
timeout 1200 python -m pytest tests/test_gpu_batch_grouped.py tests/test_gpu_edge_cases.py tests/test_gpu_alternate_paths.py tests/test_gpu_baseline_configs.py -x -q > gpurun_out/g2/pytest.log 2>&1; echo "exit $?" >> gpurun_out/g2/pytest.log
tail -15 gpurun_out/g2/pytest.log
SC_GROUP_TRACE=1 timeout 600 python tools/group_only.py 16 > gpurun_out/g2/probe.log 2> gpurun_out/g2/trace.log
cat gpurun_out/g2/probe.log
grep -c "latched at" gpurun_out/g2/trace.log
grep "latched" gpurun_out/g2/trace.log | tail -5
timeout 600 python tools/batch_group_probe.py 8 16 > gpurun_out/g2/probe2.log 2>&1
cat gpurun_out/g2/probe2.log
