for A in "--dims 96 32 128 --cin 48 --cout 32 --kernel 3 3 3" "--dims 192 64 128 --cin 64 --cout 32" "--dims 192 64 128 --cin 16 --cout 32" "--cin 32 --cout 32"; do
  echo "== $A"
  python tools/bench_igemm.py $A --reps 10 2>&1 | tail -1
  for CK in 8 16 48; do
    python tools/bench_igemm.py $A --reps 10 --ck $CK 2>&1 | tail -1
  done
done
