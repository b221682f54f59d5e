cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "conv or 512 or epilogue or igemm" > gpurun_out/ops_d.log 2>&1; tail -2 gpurun_out/ops_d.log
for A in "--depth 1 --ck 16 --mtw 2" "--depth 1 --ck 8 --mtw 2" "--depth 1 --ck 16 --tile 4 8 8" "--depth 1 --ck 16 --tile 8 8 8" "--depth 1 --ck 8 --tile 8 8 8" "--depth 2 --ck 8 --tile 4 8 8" "--depth 1 --ck 16 --tile 4 8 8 --stats"; do
  echo -n "96->48 $A : "; timeout 120 python tools/bench_igemm.py --dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 $A --reps 10 2>&1 | tail -1
  echo -n "48->48 $A : "; timeout 120 python tools/bench_igemm.py --dims 96 32 128 --cin 48 --cout 48 --kernel 3 3 3 $A --reps 10 2>&1 | tail -1
done
