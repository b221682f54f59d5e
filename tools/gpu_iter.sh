cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_network.py -q -x -p no:cacheprovider -k "graph_replay or adam_step or eval_cache or dropout_mask" > gpurun_out/g1.log 2>&1; tail -5 gpurun_out/g1.log
for G in 0 1; do echo "== VSSEG_GRAPHS=$G"; VSSEG_GRAPHS=$G python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2> gpurun_out/bg$G.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','loss')}); print(d['sliding_window'])"; done
