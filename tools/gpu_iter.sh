cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused_parity or streaming_kernel" -p no:cacheprovider 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_benchmark_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do timeout 300 python tools/time_step.py 30 2>&1 | tail -1; done
VSSEG_PROFILE_ROWS=300 timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --swi-volumes 0 --no-parity --profile > gpurun_out/it/b.json 2> gpurun_out/it/b.err
grep "D=-4" gpurun_out/it/b.err | cut -c1-170
