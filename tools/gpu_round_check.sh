# full GPU check of the round: test suite, then a bench run that re-measures every launch plan and stores the choices
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/r2_t.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_t.log
export VSSEG_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_r2.json
rm -f $VSSEG_TUNE_CACHE
VSSEG_AUTOTUNE=force timeout 1500 python bench.py --steps 10 --warmup 3 --profile --no-cpu-baseline > gpurun_out/r2_b.log 2> gpurun_out/r2_b.err
echo "bench rc=$?" >> gpurun_out/r2_b.err
tail -8 gpurun_out/r2_t.log; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_b.log").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "conv_stack_mfma_frac", "loss")}); print(d["parity"]); print(d["sliding_window"])
PY
