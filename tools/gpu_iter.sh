cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "streaming" -p no:cacheprovider 2>&1 | tail -5
for p in 3 4 5; do echo "== per CU $p"; VSSEG_SCONV_PERCU=$p timeout 300 python tools/bench_sconv.py 2>&1 | grep streaming | sed 's/(. GB.s, /(/; s/ (. GB.s)//'; done > gpurun_out/sconv_sweep.log 2>&1
cat gpurun_out/sconv_sweep.log
