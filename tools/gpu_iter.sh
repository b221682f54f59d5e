cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "bn_dropout or dropout" -p no:cacheprovider 2>&1 | tail -1
timeout 300 python tools/bench_eltwise.py --c 16 32 2>&1 | grep "bn_act_fwd\|bwd_reduce  \|---"
for i in 1 2; do timeout 300 python tools/time_step.py 30 2>&1 | tail -1; done
