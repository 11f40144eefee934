python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
SC_EIG_TRACE=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/b10.err > gpurun_out/bench10.json; grep jacobi gpurun_out/b10.err | tail -1;  python -c "
import json; d=json.loads(open('gpurun_out/bench10.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'], d['parity'], d['eig'])"
python tools/bench_configs.py 2>/dev/null | grep -E "cfg2_ms|cfg4|cfg5_utt|single_n[0-9]+_ms|\"eig\"" 
