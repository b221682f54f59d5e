cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
python -m pytest tests/test_gpu_ops.py -q -k "compute_weight_gradient" 2>&1 | tail -2
for x in 2 96; do python tools/bench_wgrad.py --dims $x 32 128 --cin 96 --cout 48 --kernel 3 3 3 --compute-only 2>&1 | grep "cg="; done
python tools/bench_wgrad.py --dims 48 16 64 --cin 128 --cout 64 --kernel 3 3 3 --compute-only 2>&1 | grep "cg="
for i in 1 2 3; do
VSSEG_COMPUTE_WGRAD=0 python tools/time_step.py 30 2>&1 | tail -1 | sed "s/^/cw=0 /"
VSSEG_COMPUTE_WGRAD=1 python tools/time_step.py 30 2>&1 | tail -1 | sed "s/^/cw=1 /"
done
