# usage: bash tools/sweep_depth0.sh -> double-buffered prefetch (depth 1) vs single buffer / more resident workgroups (depth -1)
for A in "--cin 16 --cout 16" "--cin 32 --cout 16" "--cin 32 --cout 16 --ck 16" "--cin 32 --cout 2" "--dims 192 64 128 --cin 64 --cout 32" "--dims 192 64 128 --cin 64 --cout 32 --ck 16" "--dims 192 64 128 --cin 32 --cout 32" "--dims 96 32 128 --cin 96 --cout 48 --kernel 3 3 3"; do
  echo "== $A"
  for D in 1 -1; do python tools/bench_igemm.py $A --reps 10 --depth $D 2>&1 | tail -1; done
done
