cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_benchmark_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for cfg in "VSSEG_CLASS_INTERLEAVE=0" "VSSEG_CLASS_INTERLEAVE=1" "VSSEG_CLASS_INTERLEAVE=0" "VSSEG_CLASS_INTERLEAVE=1"; do echo $cfg; env $cfg timeout 300 python tools/time_step.py 30 2>&1 | tail -1; done
