# usage: bash tools/sweep_tile2.sh -> tile shapes again, now that occupancy is no longer the limiter (single-buffer mode)
for A in "--cin 32 --cout 16" "--cin 16 --cout 16" "--dims 192 64 128 --cin 64 --cout 32 --ck 32"; do
  echo "== $A"
  for T in "8 8 4" "8 4 8" "4 8 8" "4 4 16" "16 4 4" "4 16 4" "2 8 16"; do
    python tools/bench_igemm.py $A --reps 10 --depth -1 --tile $T 2>&1 | tail -1
  done
done
