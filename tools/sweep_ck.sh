# usage: bash tools/sweep_ck.sh -> igemm time vs channel-chunk size (LDS footprint / resident workgroups trade-off; tuning tool)
for A in "--cin 32 --cout 16" "--cin 16 --cout 32 --kind conv_dgrad" "--dims 192 64 128 --cin 64 --cout 32" "--dims 192 64 128 --cin 32 --cout 64 --kind conv_dgrad" "--dims 192 64 128 --cin 32 --cout 32"; do
  echo "== $A"
  for CK in 8 16 32 64; do
    python tools/bench_igemm.py $A --reps 10 --ck $CK 2>&1 | tail -1
  done
done
