cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "bn_dropout or wgrad_narrow or dropout" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/bench_eltwise.py --c 16 2>&1 | tail -12
VSSEG_PROFILE_ROWS=10 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --swi-volumes 0 --no-parity --profile > gpurun_out/it/b.json 2> gpurun_out/it/b.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/it/b.json").read().strip().splitlines()[-1])
fam = {}
for r in d["roofline_table"]:
    k = r["kernel"].split("<")[0]
    fam[k] = fam.get(k, 0) + r["ms"]
print("bench", round(d["ms_per_step"], 3), "ms/step", {k: round(v, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:11]})
PY
