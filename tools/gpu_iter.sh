# one GPU iteration: full test suite, then a forced re-measurement of every launch plan (training + sliding-window plans) with the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/it/t.log 2>&1; echo "pytest rc=$?" >> gpurun_out/it/t.log
tail -4 gpurun_out/it/t.log
export VSSEG_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/it/tune.json
rm -f $VSSEG_TUNE_CACHE
VSSEG_AUTOTUNE=force timeout 1500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/it/b.json 2> gpurun_out/it/b.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/it/b.json").read().strip().splitlines()[-1])
print("retuned", round(d["ms_per_step"], 3), "ms/step", d["value"], d["parity"]["pass"], d["sliding_window"])
PY
