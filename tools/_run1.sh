cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "compute_weight_gradient" 2>&1 | tail -15 > gpurun_out/t1.log
rm -f gpurun_out/b1.log
for e in 0 1 2 3; do echo "EXP $e" >> gpurun_out/b1.log; VSSEG_CW_EXP=$e timeout 300 python tools/bench_wgrad.py --dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 --compute-only >> gpurun_out/b1.log 2>&1; done
cat gpurun_out/t1.log gpurun_out/b1.log
