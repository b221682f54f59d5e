# Re-measures the launches the deep-level kernel (csrc/dconv.hip, depth -7) is offered for into gpurun_out/r5a/tune.json, then A/Bs the training step and the sliding
# window with and without it (VSSEG_DEEP=0 keeps it out; that arm runs WITHOUT a writable cache: the first version of this script let it overwrite the measured choices):
#   bash tools/tune_deep.sh
mkdir -p gpurun_out/r5a
rm -f gpurun_out/r5a/tune.json
VSSEG_TUNE_CACHE=$PWD/gpurun_out/r5a/tune.json VSSEG_RETUNE_DEPTHS=-7 python tools/tune_shapes.py 4x384x128x128 1x384x128x128 1x384x384x64 > gpurun_out/r5a/tune.log 2>&1
cp gpurun_out/r5a/tune.json /tmp/tune_ro.json
for i in 1 2 3; do
VSSEG_DEEP=0 python tools/time_step.py 30 2>&1 | tail -1 | sed "s/^/deep=0 /"
VSSEG_TUNE_CACHE=/tmp/tune_ro.json VSSEG_DEEP=1 python tools/time_step.py 30 2>&1 | tail -1 | sed "s/^/deep=1 /"
done > gpurun_out/r5a/ab.txt 2>&1
VSSEG_DEEP=0 python tools/time_swi.py > gpurun_out/r5a/swi0.txt 2>&1
VSSEG_TUNE_CACHE=/tmp/tune_ro.json VSSEG_DEEP=1 python tools/time_swi.py > gpurun_out/r5a/swi1.txt 2>&1
tail -3 gpurun_out/r5a/tune.log; cat gpurun_out/r5a/ab.txt; sed -n 2,4p gpurun_out/r5a/swi0.txt; sed -n 2,4p gpurun_out/r5a/swi1.txt
