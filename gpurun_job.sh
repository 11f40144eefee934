python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/b13.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['concurrent_streams'], d['roofline'])"; tail -3 gpurun_out/b13.err
