cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "bn_dropout or fused_bn" -p no:cacheprovider 2>&1 | tail -1
timeout 300 python tools/bench_eltwise.py --c 16 32 48 64 96 2>&1 | grep "reduce keep" | tr -s ' ' | cut -d' ' -f4-7 | paste -sd'|'
for b in 400000 65536 400000 65536; do echo -n "budget=$b: "; VSSEG_BN_REDUCE_ATOMICS=$b timeout 300 python tools/time_step.py 30 2>&1 | tail -1; done
