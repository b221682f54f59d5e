cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused_bn or streaming_kernel_equals or compute_kernel_equals" -p no:cacheprovider 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_benchmark_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for cfg in "VSSEG_BNRED=0" "VSSEG_BNRED=1" "VSSEG_BNRED=0" "VSSEG_BNRED=1"; do echo $cfg; env $cfg timeout 300 python tools/time_step.py 30 2>&1 | tail -1; done
