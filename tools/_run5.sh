cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
for i in 1 2; do
VSSEG_NARROW_FWD=0 python tools/time_step.py 30 2>&1 | tail -1 | sed "s/^/nf=0 /"
VSSEG_NARROW_FWD=1 python tools/time_step.py 30 2>&1 | tail -1 | sed "s/^/nf=1 /"
done > $O/ab.txt 2>&1
VSSEG_NARROW_FWD=0 python tools/time_swi.py 10 2>&1 | grep -v amdgpu | sed "s/^/nf=0 /" | head -4 > $O/swi.txt
VSSEG_NARROW_FWD=1 python tools/time_swi.py 10 2>&1 | grep -v amdgpu | sed "s/^/nf=1 /" | head -4 >> $O/swi.txt
cat $O/ab.txt $O/swi.txt
python -m pytest tests/test_gpu_network.py tests/test_gpu_benchmark_parity.py -x -q 2>&1 | tail -3
