cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_benchmark_parity.py tests/test_gpu_network.py -q -s -k "golden or c3 or plans" 2>&1 | grep -v "^  \|Warning\|amdgpu" | grep "parity\|C3\|fp32 b1\|passed\|failed\|Error\|assert" | cut -c1-1500 > gpurun_out/t4.log
cat gpurun_out/t4.log
