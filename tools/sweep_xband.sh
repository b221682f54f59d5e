# usage: bash tools/sweep_xband.sh  -> igemm time of the HBM-bound layers for several tile-order band widths (tuning tool)
for XB in 1 2 4 8 16; do
  echo "== VSSEG_TILE_XBAND=$XB"
  for A in "--cin 16 --cout 16" "--cin 32 --cout 16" "--cin 16 --cout 32 --kind conv_dgrad" "--dims 192 64 128 --cin 64 --cout 32" "--dims 192 64 128 --cin 32 --cout 32"; do
    VSSEG_TILE_XBAND=$XB python tools/bench_igemm.py $A --reps 10 2>&1 | tail -1
  done
done
