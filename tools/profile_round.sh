# Collects the round's evidence on the GPU box: bench line (with CPU baseline), rocprofv3 kernel trace + PMC HBM passes.
# usage (from the repo root on the GPU box):  bash tools/profile_round.sh r01
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
# the first run measures the launch plans and stores its choices; the profiled runs replay exactly those plans
export VSSEG_TUNE_CACHE=$OUT/tuned_plans.json
VSSEG_AUTOTUNE=force python $R/bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err   # re-measures every launch plan
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py --steps 5 --warmup 2 --swi-volumes 0 --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -- python $R/bench.py --steps 2 --warmup 1 --swi-volumes 0 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -- python $R/bench.py --steps 2 --warmup 1 --swi-volumes 0 --no-cpu-baseline > $OUT/write.log 2>&1
cd $R
python tools/rocprof_summary.py kernel $OUT/kt/*/*.db > $OUT/kernel_stats.txt
python tools/rocprof_summary.py pmc $OUT/fetch/*/*.db $OUT/write/*/*.db $OUT/roofline_traffic.json > $OUT/pmc_hbm.txt
rm -rf $OUT/kt $OUT/fetch $OUT/write
tail -1 $OUT/bench.json | cut -c1-400; head -12 $OUT/kernel_stats.txt; head -8 $OUT/pmc_hbm.txt
