# usage: bash tools/sweep_depth.sh  -> igemm time of the HBM-bound layers for DMA ring depths 1..3 (tuning tool)
for D in 1 2 3; do
  echo "== depth $D"
  for A in "--cin 16 --cout 16" "--cin 32 --cout 16" "--cin 16 --cout 16 --stats" "--dims 192 64 128 --cin 64 --cout 32" "--dims 192 64 128 --cin 32 --cout 32" "--dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3"; do
    python tools/bench_igemm.py $A --reps 10 --depth $D 2>&1 | tail -1
  done
done
