cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/bench_sconv.py 2>&1 | grep streaming | head -3
