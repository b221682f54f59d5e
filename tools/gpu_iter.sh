cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/it/t.log 2>&1; echo "pytest rc=$?" >> gpurun_out/it/t.log
tail -3 gpurun_out/it/t.log
export VSSEG_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/it/tune.json
rm -f $VSSEG_TUNE_CACHE
VSSEG_PROFILE_ROWS=60 VSSEG_AUTOTUNE=force timeout 1500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile > gpurun_out/it/b.json 2> gpurun_out/it/b.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/it/b.json").read().strip().splitlines()[-1])
fam = {}
for r in d["roofline_table"]:
    k = r["kernel"].split("<")[0]
    fam[k] = fam.get(k, 0) + r["ms"]
print("retuned", round(d["ms_per_step"], 3), "ms/step", d["parity"]["pass"], {k: round(v, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:11]})
print(d["sliding_window"]["volumes_per_sec"], d["sliding_window"]["sw_batch_size_4"])
PY
grep "wgrad<bf16,[345]>" gpurun_out/it/b.err | head -12 | cut -c1-220
