python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -15
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/b12.err > gpurun_out/bench12.json;  python -c "
import json; d=json.loads(open('gpurun_out/bench12.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'], d['parity'], d['eig'])"
