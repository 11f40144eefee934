cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys
j=json.loads(sys.stdin.read())
print(round(j['value'],2), [(k['kernel'][:12], round(k['us'],1)) for k in j['roofline']['kernels']], round(j['stage_ms']['refine'],4))"; }
for i in 1 2; do
echo new; timeout 600 python bench.py --no-extras --no-cpu-baseline | show
echo oldblur; SPECTRALCLUSTER_AMD_LIB=$PWD/gpurun_alt/libalt_oldblur.so timeout 600 python bench.py --no-extras --no-cpu-baseline | show
done
