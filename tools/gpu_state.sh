# state of the tree on the GPU box: test suite, default bench line, rocprofv3 kernel trace of a short bench run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/state
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/state/t.log 2>&1
echo "pytest rc=$?" >> gpurun_out/state/t.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/state/bench.json 2> gpurun_out/state/bench.err
echo "bench rc=$?" >> gpurun_out/state/bench.err
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/state; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py --steps 5 --warmup 2 --swi-volumes 0 --no-cpu-baseline --no-parity > $OUT/kt.log 2>&1
cd $R
python tools/rocprof_summary.py kernel $OUT/kt/*/*.db > $OUT/kernel_stats.txt
rm -rf $OUT/kt
tail -4 $OUT/t.log; tail -1 $OUT/bench.json | cut -c1-400; head -30 $OUT/kernel_stats.txt
