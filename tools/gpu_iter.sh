cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "512 or h_chunk or wgrad" > gpurun_out/ops_b.log 2>&1; tail -15 gpurun_out/ops_b.log
