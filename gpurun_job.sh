python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -2
SC_EIG_TRACE=1 python tools/stage_probe.py predict300 300 3 2>&1 | grep jacobi | tail -2
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms'], d['parity'])"
python tools/bench_configs.py 2>/dev/null | grep -E "cfg2_ms|cfg4|cfg5_utt|single_n[0-9]+_ms" 
