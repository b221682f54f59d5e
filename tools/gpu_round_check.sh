cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/r2_t1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_t1.log
export VSSEG_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_r2a.json
timeout 900 python bench.py --steps 10 --warmup 3 --profile --no-cpu-baseline > gpurun_out/r2_b1.log 2> gpurun_out/r2_b1.err
echo "bench rc=$?" >> gpurun_out/r2_b1.err
tail -30 gpurun_out/r2_t1.log; tail -c 1500 gpurun_out/r2_b1.log
