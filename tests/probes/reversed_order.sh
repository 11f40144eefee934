# The GPU suite in REVERSED collection order (state a test leaves on the default handle -- arenas,
# resident buffers, flag words, switches read once -- must not matter to the next one):
#   gpurun -- bash tests/probes/reversed_order.sh > profiles/rNN_gpu_tests_reversed_order.txt
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu --co -q 2>/dev/null | grep "::" | tac > /tmp/rev_ids.txt
wc -l < /tmp/rev_ids.txt
timeout 1500 python -m pytest -q -p no:cacheprovider @/tmp/rev_ids.txt 2>&1 | tail -5
