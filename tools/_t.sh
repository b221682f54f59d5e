rm -f gpurun_out/tune_c5.json
VSSEG_TUNE_CACHE=$PWD/gpurun_out/tune_c5.json VSSEG_RETUNE_DEPTHS=-8,-9 python tools/tune_shapes.py 1x384x384x64 2x384x128x128 2>&1 | tail -3
