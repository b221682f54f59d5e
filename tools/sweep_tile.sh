# usage: bash tools/sweep_tile.sh  -> igemm time of the HBM-bound layers for several tile shapes (tuning tool)
for T in "8 8 4" "8 4 8" "4 8 8" "4 4 16" "2 8 16" "8 2 16" "4 2 32" "2 4 32" "2 2 64" "16 4 4" "4 16 4"; do
  echo "== tile $T"
  for A in "--cin 16 --cout 16" "--cin 32 --cout 16" "--dims 192 64 128 --cin 32 --cout 32"; do
    python tools/bench_igemm.py $A --reps 10 --tile $T 2>&1 | tail -1
  done
done
