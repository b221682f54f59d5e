# occupancy scaling of the HBM-bound implicit-GEMM layers: time vs resident workgroups per CU (tuning experiment)
cd $GRAFT_REPO_ROOT
for A in "--cin 16 --cout 16 --stats --depth -1" "--cin 16 --cout 16 --depth -1" "--cin 32 --cout 16 --depth -1" "--cin 32 --cout 16 --depth -1 --ck 16" "--cin 16 --cout 16 --stats --depth 1" "--cin 16 --cout 16 --stats --depth 3" "--dims 192 64 128 --cin 64 --cout 32 --depth -1 --ck 32" "--dims 192 64 128 --cin 32 --cout 32 --depth -1"; do
  for N in 1 2 3 4 5 6; do
    echo -n "percu<=$N $A : "; VSSEG_IG_PERCU=$N python tools/bench_igemm.py $A --reps 10 2>&1 | tail -1
  done
done
