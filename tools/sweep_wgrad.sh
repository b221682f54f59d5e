cd $GRAFT_REPO_ROOT
for cfg in "--dims 192 64 128 --cin 64 --cout 32" "--dims 384 128 128 --cin 32 --cout 16" "--dims 384 128 128 --cin 16 --cout 16" "--dims 192 64 128 --cin 32 --cout 32" "--dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3" "--dims 384 128 128 --cin 32 --cout 2" "--dims 384 128 128 --cin 16 --cout 1"; do
  python tools/bench_wgrad.py $cfg --blocks 128 256 512 1024
done > gpurun_out/wg_sweep1.log 2>&1
tail -80 gpurun_out/wg_sweep1.log
