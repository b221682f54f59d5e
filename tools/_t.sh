for r in 1 2; do
for t in 0 1; do
  echo "== transition=$t"; VSSEG_TRANSITION=$t python tools/time_step.py 30 2>&1 | tail -1
  VSSEG_TRANSITION=$t python tools/time_swi.py 10 2>&1 | grep "segmentation_predictor" | head -2
done; done
