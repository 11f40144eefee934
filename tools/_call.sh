cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys
j=json.loads(sys.stdin.read())
print(round(j['value'],2), round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['stage_ms'].items() if v}, [round(k['us'],1) for k in j['roofline']['kernels']])"; }
for i in 1 2; do
echo nofence; timeout 600 python bench.py --no-extras --no-concurrent --no-cpu-baseline | show
echo fence; SC_EVENT_SYSTEM_FENCE=1 timeout 600 python bench.py --no-extras --no-concurrent --no-cpu-baseline | show
done
timeout 300 python tools/single_sizes.py 500 1650 3000
