cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
rm -f $O/tune.json
VSSEG_TUNE_CACHE=$PWD/$O/tune.json python tools/tune_shapes.py 4x384x128x128 > $O/tune.log 2>&1
tail -2 $O/tune.log
cp $O/tune.json /tmp/tune_ro.json
for i in 1 2 3; do
VSSEG_COMPUTE_WGRAD=0 python tools/time_step.py 30 2>&1 | tail -1 | sed "s/^/cw=0 /"
VSSEG_TUNE_CACHE=/tmp/tune_ro.json VSSEG_COMPUTE_WGRAD=1 python tools/time_step.py 30 2>&1 | tail -1 | sed "s/^/cw=1 /"
done > $O/ab.txt 2>&1
cat $O/ab.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6a/tune.json"))
for k, v in d.items():
    if k.startswith("wgrad4"):
        print(k, v)
PY
