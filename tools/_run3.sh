cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
rm -f $O/tune.json
VSSEG_TUNE_CACHE=$PWD/$O/tune.json python tools/tune_shapes.py 4x384x128x128 1x384x128x128 2x384x128x128 1x384x384x64 2x384x384x64 > $O/tune.log 2>&1
grep -v amdgpu $O/tune.log | tail -5
