# Re-measures the launches the deep-level kernel (csrc/dconv.hip, depth -7) is offered for into gpurun_out/r5a/tune.json and A/Bs VSSEG_DEEP=0/1 (training step, sliding window): bash tools/tune_deep.sh
mkdir -p gpurun_out/r5a
export VSSEG_TUNE_CACHE=$PWD/gpurun_out/r5a/tune.json VSSEG_RETUNE_DEPTHS=-7
python tools/tune_shapes.py 4x384x128x128 1x384x128x128 > gpurun_out/r5a/tune.log 2>&1
unset VSSEG_RETUNE_DEPTHS
for i in 1 2; do
VSSEG_DEEP=0 VSSEG_OVERLAP=1 python tools/time_step.py 20 2>&1 | tail -1
VSSEG_DEEP=1 VSSEG_OVERLAP=1 python tools/time_step.py 20 2>&1 | tail -1
done > gpurun_out/r5a/ab.txt 2>&1
VSSEG_DEEP=0 python tools/time_swi.py > gpurun_out/r5a/swi0.txt 2>&1
VSSEG_DEEP=1 python tools/time_swi.py > gpurun_out/r5a/swi1.txt 2>&1
tail -3 gpurun_out/r5a/tune.log; cat gpurun_out/r5a/ab.txt; tail -3 gpurun_out/r5a/swi0.txt; tail -3 gpurun_out/r5a/swi1.txt
