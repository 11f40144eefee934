cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_autotune_sweep.py -x -q 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_predict.py -x -q -k "autotune or config4 or AutoTune" 2>&1 | tail -5
timeout 600 python bench.py --workload autotune16 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print(json.dumps(j['autotune16']))"
SC_SWEEP_ONE_BY_ONE=1 timeout 600 python bench.py --workload autotune16 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print(json.dumps(j['autotune16']))"
