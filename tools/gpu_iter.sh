# one GPU iteration of the round: test suite + A/B bench runs of the switches under test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/it/t.log 2>&1; echo "pytest rc=$?" >> gpurun_out/it/t.log
tail -3 gpurun_out/it/t.log
timeout 300 python tools/bench_eltwise.py --c 16 32 2>&1 | tail -24
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --swi-volumes 0 --no-parity"
for cfg in "" "VSSEG_WGRAD_WALK=0" "VSSEG_KEEPMASK=0"; do
  env $cfg timeout 600 $B > gpurun_out/it/b.json 2> gpurun_out/it/b.err
  python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/it/b.json").read().strip().splitlines()[-1])
fam = {}
for r in d["roofline_table"]:
    k = r["kernel"].split("<")[0]
    fam[k] = fam.get(k, 0) + r["ms"]
print(sys.argv[1] or "default", round(d["ms_per_step"], 3), "ms/step |", {k: round(v, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:9]})
PY
done
