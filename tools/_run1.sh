cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "compute_weight_gradient" 2>&1 | tail -3 > gpurun_out/t1.log
rm -f gpurun_out/b1.log
for e in 0 1 6 7 0; do echo "EXP $e" >> gpurun_out/b1.log; VSSEG_CW_EXP=$e timeout 300 python tools/bench_wgrad.py --dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 --compute-only 2>&1 | grep "cg=" >> gpurun_out/b1.log; done
for cfg in "96 32 128 48 48" "96 32 128 32 48" "48 16 64 128 64" "48 16 64 64 64" "48 16 64 48 64"; do set -- $cfg; timeout 300 python tools/bench_wgrad.py --dims $1 $2 $3 --cin $4 --cout $5 --kernel 3 3 3 --compute-only 2>&1 | grep -v "amdgpu.ids\|rejected\|sb=0" >> gpurun_out/b1.log; done
cat gpurun_out/t1.log gpurun_out/b1.log
