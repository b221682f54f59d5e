# Collects the round's evidence on the GPU box: bench line (with CPU baseline), rocprofv3 kernel traces of the training steps AND of the sliding-window
# path, PMC passes (HBM bytes; MFMA / LDS / issue counters).   usage (from the repo root on the GPU box):  bash tools/profile_round.sh r03
# Counters are collected in their own runs (--pmc only, never combined with a trace).  The launch plans are the shipped ones (vs_seg_amd/tuned_gfx950.json).
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
TRAIN="--swi-volumes 0 --swi-cases 0 --fp32-steps 0 --no-cpu-baseline --no-parity"
# HBM traffic first: bench.py reads profiles/roofline_traffic.json, so the committed bench line and the traffic table come from the SAME build and run
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -- python $R/bench.py --steps 2 --warmup 1 $TRAIN > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -- python $R/bench.py --steps 2 --warmup 1 $TRAIN > $OUT/write.log 2>&1
(cd $R && python tools/rocprof_summary.py pmc $OUT/fetch/*/*.db $OUT/write/*/*.db $OUT/roofline_traffic.json > $OUT/pmc_hbm.txt && cp $OUT/roofline_traffic.json profiles/roofline_traffic.json)
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
# kernel trace of the training steps with ONE stream and eager launches (VSSEG_OVERLAP=0 VSSEG_GRAPHS=0): no kernel shares the GPU with a concurrent weight gradient,
# so the per-kernel durations of kernel_stats.txt are clean (the product's default schedule — two streams, hipGraph replay — is what bench.json times)
VSSEG_OVERLAP=0 VSSEG_GRAPHS=0 rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py --steps 5 --warmup 2 $TRAIN > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kts -- python $R/tools/profile_eval.py 8 > $OUT/kts.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/sfetch -- python $R/tools/profile_eval.py 2 > $OUT/sfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/swrite -- python $R/tools/profile_eval.py 2 > $OUT/swrite.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/sq1 -- python $R/bench.py --steps 2 --warmup 1 $TRAIN > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq2 -- python $R/bench.py --steps 2 --warmup 1 $TRAIN > $OUT/sq2.log 2>&1
# the product's DEFAULT schedule (two streams, hipGraph replay of the forward) traced once more for the host-side question: how long is the GPU idle inside a step
# (tools/trace_gaps.py), and the true start / end of every kernel of one step (tools/step_timeline.py)
rocprofv3 --kernel-trace -d $OUT/ktd -- python $R/tools/time_step.py 8 > $OUT/ktd.log 2>&1
cd $R
python tools/trace_gaps.py $OUT/ktd/*/*.db > $OUT/gaps.txt 2>&1
python tools/step_timeline.py $OUT/ktd/*/*.db > $OUT/step_timeline.txt 2> /dev/null
python - "$OUT" <<'PYEOF'
import json, re, sys
out = sys.argv[1]
line = open(out + "/bench.json").read().strip().splitlines()[-1]
d = json.loads(line)
m = re.search(r"wall ([\d.]+) ms per step, GPU busy ([\d.]+) ms, idle ([\d.]+) ms in (\d+) gaps", open(out + "/gaps.txt").read())
if m:
    d["gpu_idle_in_step"] = dict(wall_ms=float(m.group(1)), busy_ms=float(m.group(2)), idle_ms=float(m.group(3)), idle_share=float(m.group(3)) / float(m.group(1)), gaps_per_step=int(m.group(4)),
                                 source="rocprofv3 --kernel-trace of tools/time_step.py under the default schedule (two streams, eager backward), union of the kernel intervals: tools/trace_gaps.py")
open(out + "/bench.json", "w").write(json.dumps(d) + "\n")
PYEOF
rm -rf $OUT/ktd
# BASELINE config 5 at its full case count, once per round (the default bench line carries 8 cases)
python bench.py --steps 2 --warmup 1 --swi-volumes 0 --fp32-steps 0 --no-cpu-baseline --no-parity --swi-cases 242 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['sharded_cases']))" > $OUT/sharded_242.json
# what a dependent stage boundary costs (launch / hipGraph node / in-kernel grid barrier) and the deep-level kernel against the other plans, launch by launch
(tools/probes/chain_probe > $OUT/chain_probe.txt 2>&1 || true)
(tools/probes/fork_probe > $OUT/fork_probe.txt 2>&1 || true)  # what a fork of the side stream costs the main one: event record against an event bound to the kernel
(timeout 600 python tools/bench_dconv.py 4 > $OUT/dconv_bench.txt 2>&1 || true)
(timeout 300 python tools/bench_chain.py 1 2>&1 | grep -v amdgpu.ids > $OUT/chain_bench.txt || true)
# round 6: the compute weight-gradient kernel against the tile kernel on the six 3x3x3 stride-1 layers of levels 2-3, and the narrow-output convolution
(for cfg in "96 32 128 96 48" "96 32 128 48 48" "96 32 128 32 48" "48 16 64 128 64" "48 16 64 64 64" "48 16 64 48 64"; do set -- $cfg; timeout 300 python tools/bench_wgrad.py --dims $1 $2 $3 --cin $4 --cout $5 --kernel 3 3 3 2>&1 | grep -v "amdgpu.ids\|rejected"; done > $OUT/cwgrad_bench.txt || true)
(timeout 300 python tools/bench_cconv.py --compute-only --batch 4 2>&1 | grep -v amdgpu.ids > $OUT/cconv_bench.txt; timeout 300 python tools/bench_cconv.py --compute-only --batch 1 2>&1 | grep -v amdgpu.ids >> $OUT/cconv_bench.txt || true)
(timeout 600 python tools/bench_tconv.py 4 2>&1 | grep -v amdgpu.ids > $OUT/tconv_bench.txt; timeout 300 python tools/bench_tconv.py 1 2>&1 | grep -v amdgpu.ids | grep -E "^==|transition" >> $OUT/tconv_bench.txt || true)
(timeout 600 python tools/bench_gconv.py 4 2>&1 | grep -v amdgpu.ids > $OUT/gconv_bench.txt; timeout 300 python tools/bench_gconv.py 1 2>&1 | grep -v amdgpu.ids | grep -E "^==|D=-9" >> $OUT/gconv_bench.txt || true)
(timeout 300 python tools/bench_nconv.py 4 2>&1 | grep -v amdgpu.ids > $OUT/nconv_bench.txt; timeout 300 python tools/bench_nconv.py 1 2>&1 | grep -v amdgpu.ids >> $OUT/nconv_bench.txt || true)
(cd /tmp; B="python $R/tools/bench_wgrad.py --dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 --compute-only --reps 3"
 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/cwa -- $B > /dev/null 2>&1
 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES -d $OUT/cwb -- $B > /dev/null 2>&1
 rocprofv3 --pmc FETCH_SIZE -d $OUT/cwc -- $B > /dev/null 2>&1
 rocprofv3 --pmc WRITE_SIZE -d $OUT/cwd -- $B > /dev/null 2>&1
 cd $R; python tools/pmc_dump.py "cwgrad_kernel<3, 1, 2" $OUT/cwa/*/*.db $OUT/cwb/*/*.db $OUT/cwc/*/*.db $OUT/cwd/*/*.db > $OUT/cwgrad_pmc.txt; rm -rf $OUT/cwa $OUT/cwb $OUT/cwc $OUT/cwd) || true
python tools/rocprof_summary.py kernel $OUT/kt/*/*.db > $OUT/kernel_stats.txt
python tools/rocprof_summary.py kernel $OUT/kts/*/*.db > $OUT/swi_kernel_stats.txt
python tools/rocprof_summary.py pmc $OUT/sfetch/*/*.db $OUT/swrite/*/*.db > $OUT/swi_pmc_hbm.txt
python tools/rocprof_summary.py sq $OUT/sq1/*/*.db $OUT/sq2/*/*.db > $OUT/pmc_sq.txt
rm -rf $OUT/kt $OUT/kts $OUT/fetch $OUT/write $OUT/sfetch $OUT/swrite $OUT/sq1 $OUT/sq2
tail -1 $OUT/bench.json | cut -c1-300; head -14 $OUT/kernel_stats.txt; head -8 $OUT/swi_kernel_stats.txt; head -8 $OUT/pmc_hbm.txt; head -16 $OUT/pmc_sq.txt; head -3 $OUT/gaps.txt; cat $OUT/chain_probe.txt; tail -1 $OUT/dconv_bench.txt; cut -c1-300 $OUT/sharded_242.json
