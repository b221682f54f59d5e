cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "compute_weight_gradient" 2>&1 | tail -12 > gpurun_out/t1.log
rm -f gpurun_out/b1.log
for x in 2 96; do timeout 300 python tools/bench_wgrad.py --dims $x 32 128 --cin 96 --cout 48 --kernel 3 3 3 --compute-only 2>&1 | grep "cg=" >> gpurun_out/b1.log; done
python tools/_dbg.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/b1.log
cat gpurun_out/t1.log gpurun_out/b1.log
