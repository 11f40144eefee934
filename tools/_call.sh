cd $GRAFT_REPO_ROOT
run() { echo "$@"; env "$@" timeout 300 python tools/group_only.py 16 | tail -1; }
run A=default
run SC_GEMM_GROUP_PERSIST=0
run A=default
run SC_GEMM_GROUP_PERSIST=0
timeout 600 python -m pytest tests/test_gpu_batch_grouped.py -x -q 2>&1 | tail -3
