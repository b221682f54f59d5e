# one GPU iteration of the round: full test suite, then the step time as a user runs it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/it/t.log 2>&1; echo "pytest rc=$?" >> gpurun_out/it/t.log
tail -3 gpurun_out/it/t.log; grep FAILED gpurun_out/it/t.log | head -5
for i in 1 2 3; do timeout 300 python tools/time_step.py 30 2>&1 | tail -1; done
