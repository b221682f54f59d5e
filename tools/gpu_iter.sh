cd $GRAFT_REPO_ROOT
for cfg in "VSSEG_OVERLAP_MIN_LEVEL=0" "VSSEG_OVERLAP_MIN_LEVEL=1" "VSSEG_OVERLAP_MIN_LEVEL=2" "VSSEG_OVERLAP_MIN_LEVEL=3"; do echo $cfg; env $cfg timeout 300 python tools/time_step.py 30 2>&1 | tail -1; done
