# per-phase cycle breakdown of the implicit-GEMM tile loop (tuning build: make -C vs_seg_amd/csrc PROF=1)
cd $GRAFT_REPO_ROOT
export VSSEG_IG_PROF_PRINT=1 VSSEG_LIB_PATH=$GRAFT_REPO_ROOT/vs_seg_amd/libvsseg_hip_prof.so
for N in 1 4; do for A in "--cin 16 --cout 16 --stats --depth -1" "--cin 32 --cout 16 --depth -1" "--cin 16 --cout 16 --stats --depth 1" "--dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3 --depth -1 --ck 16 --mtw 2"; do
  echo "== percu<=$N $A"; VSSEG_IG_PERCU=$N python tools/bench_igemm.py $A --reps 2 2>&1 | tail -3
done; done
