cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p2; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/bench_wgrad.py --dims 2 32 128 --cin 96 --cout 48 --kernel 3 3 3 --compute-only > $O/kt.log 2>&1
cd $R; python tools/rocprof_summary.py kernel $O/kt/*/*.db | head -6
rm -rf $O/kt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/bench_wgrad.py --dims 48 16 64 --cin 128 --cout 64 --kernel 3 3 3 --compute-only > $O/kt.log 2>&1
cd $R; python tools/rocprof_summary.py kernel $O/kt/*/*.db | head -6
rm -rf $O/kt
