R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { python tools/bench_igemm.py "$@" 2>&1 | tail -1; }
{
for B in 40000 78000 110000 158000; do
echo "=== lds budget $B"
echo "32->16 331 L0:  $(run --cin 32 --cout 16 --lds-budget $B)"
echo "16->16 331 L0:  $(run --cin 16 --cout 16 --lds-budget $B --stats)"
echo "64->32 331 L1:  $(run --dims 192 64 128 --cin 64 --cout 32 --lds-budget $B)"
echo "32->32 331 L1:  $(run --dims 192 64 128 --cin 32 --cout 32 --lds-budget $B --stats)"
echo "96->48 333 L2:  $(run --dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 --lds-budget $B)"
echo "48->48 333 L2:  $(run --dims 96 32 128 --cin 48 --cout 48 --kernel 3 3 3 --lds-budget $B --stats)"
echo "128->64 333 L3: $(run --dims 48 16 64 --cin 128 --cout 64 --kernel 3 3 3 --lds-budget $B)"
done
echo "=== mtw 2"
echo "32->16 331 L0:  $(run --cin 32 --cout 16 --mtw 2)"
echo "64->32 331 L1:  $(run --dims 192 64 128 --cin 64 --cout 32 --mtw 2)"
echo "64->32 331 L1 b110k:  $(run --dims 192 64 128 --cin 64 --cout 32 --mtw 2 --lds-budget 110000)"
echo "96->48 333 L2:  $(run --dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 --mtw 2)"
echo "96->48 333 L2 b110k:  $(run --dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 --mtw 2 --lds-budget 110000)"
} > gpurun_out/sweep2.log 2>&1
cat gpurun_out/sweep2.log
