# Collects the round's evidence on the GPU box: bench line (with CPU baseline), rocprofv3 kernel trace + PMC passes (HBM bytes, MFMA / LDS / issue counters).
# usage (from the repo root on the GPU box):  bash tools/profile_round.sh r02        (VSSEG_RETUNE=1 re-measures every launch plan first)
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
if [ -n "$VSSEG_RETUNE" ]; then
  export VSSEG_TUNE_CACHE=$OUT/tuned_plans.json
  VSSEG_AUTOTUNE=force python $R/bench.py --steps 3 --warmup 1 --swi-volumes 0 --no-cpu-baseline --no-parity > $OUT/retune.log 2>&1
fi
python $R/bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py --steps 5 --warmup 2 --swi-volumes 0 --no-cpu-baseline --no-parity > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -- python $R/bench.py --steps 2 --warmup 1 --swi-volumes 0 --no-cpu-baseline --no-parity > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -- python $R/bench.py --steps 2 --warmup 1 --swi-volumes 0 --no-cpu-baseline --no-parity > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/sq1 -- python $R/bench.py --steps 2 --warmup 1 --swi-volumes 0 --no-cpu-baseline --no-parity > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq2 -- python $R/bench.py --steps 2 --warmup 1 --swi-volumes 0 --no-cpu-baseline --no-parity > $OUT/sq2.log 2>&1
cd $R
python tools/rocprof_summary.py kernel $OUT/kt/*/*.db > $OUT/kernel_stats.txt
python tools/rocprof_summary.py pmc $OUT/fetch/*/*.db $OUT/write/*/*.db $OUT/roofline_traffic.json > $OUT/pmc_hbm.txt
python tools/rocprof_summary.py sq $OUT/sq1/*/*.db $OUT/sq2/*/*.db > $OUT/pmc_sq.txt
rm -rf $OUT/kt $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
tail -1 $OUT/bench.json | cut -c1-300; head -14 $OUT/kernel_stats.txt; head -8 $OUT/pmc_hbm.txt; head -16 $OUT/pmc_sq.txt
