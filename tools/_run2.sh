cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p2; mkdir -p $O
B="python $R/tools/bench_wgrad.py --dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 --compute-only --reps 3"
for e in 0 1; do
export VSSEG_CW_EXP=$e
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/a$e -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD -d $O/b$e -- $B > $O/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL -d $O/c$e -- $B > $O/c.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES -d $O/d$e -- $B > $O/d.log 2>&1
echo "=== EXP $e" >> $O/pmc.txt
(cd $R; python tools/pmc_dump.py "cwgrad_kernel<3, 1, 2" $O/a$e/*/*.db $O/b$e/*/*.db $O/c$e/*/*.db $O/d$e/*/*.db >> $O/pmc.txt)
rm -rf $O/a$e $O/b$e $O/c$e $O/d$e
done
cat $O/pmc.txt
