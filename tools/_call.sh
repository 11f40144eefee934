cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_batch_grouped.py -x -q 2>&1 | tail -4
run() { echo "$@"; env "$@" timeout 300 python tools/group_only.py 16 | tail -1; }
run A=default
run A=default
run SC_GEMM_GROUP_PERSIST=1
