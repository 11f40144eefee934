# A/B of the grouped batch's group formation (round 6): groups of equal cost dealt longest-first
# to the lanes (default) against round 5's equal counts dealt round-robin (SC_GROUP_EQUAL_COUNT=1):
# config 5 throughput and the projected 8-GPU speed-up (bench.py --workload batch512).
#   gpurun -- bash tests/probes/group_balance_ab.sh > profiles/rNN_group_balance_ab.txt
cd $GRAFT_REPO_ROOT
for mode in ${MODES:-cost count cost count}; do
  if [ $mode = count ]; then export SC_GROUP_EQUAL_COUNT=1; else unset SC_GROUP_EQUAL_COUNT; fi
  echo "== groups by $mode"
  GROUP_ONLY_PASSES=4 timeout 300 python tests/probes/group_only.py 16 2>&1 | tail -3
  timeout 400 python bench.py --workload batch512 --no-cpu-baseline --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['batch512']
print('bench batch512: %.0f utterances/s;' % b['value'], 'projected', {k:(round(v['speedup'],3), round(v['imbalance'],3), round(1e3*v['max_share_s'],2)) for k,v in b['projected'].items() if k in '248'})
print('   world 8 shares (ms):', b['projected']['8']['share_ms'])
"
done
