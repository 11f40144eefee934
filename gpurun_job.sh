python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench8.json 2> gpurun_out/bench8.err; python -c "
import json; d=json.loads(open('gpurun_out/bench8.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'], d['parity'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
