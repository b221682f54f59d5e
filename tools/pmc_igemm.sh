# usage: bash tools/pmc_igemm.sh "<bench_igemm args>"   -> prints per-kernel averages of several PMC sets (tuning tool)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
ARGS="$1"
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" ; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $SET -d /tmp/pmc_out -- python $R/tools/bench_igemm.py $ARGS --reps 3 > /tmp/pmc.log 2>&1
  python - <<PY
import sqlite3, glob
f = glob.glob("/tmp/pmc_out/*/*.db")
if not f:
    print("no db; log tail:"); print(open("/tmp/pmc.log").read()[-1500:])
else:
    cur = sqlite3.connect(f[0]).cursor()
    rows = list(cur.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%igemm%' group by counter_name"))
    for r in rows: print(f"{r[0]:28s} n={r[1]:3d} avg={r[2]:16.1f}")
PY
done
