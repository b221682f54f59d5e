cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/it
VSSEG_PROFILE_ROWS=300 timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --swi-volumes 0 --no-parity --profile > gpurun_out/it/b.json 2> gpurun_out/it/b.err
grep "D=-4\|convT_fwd\|conv_dgrad q=(192, 64, 128) K=16x[1-4] \|conv_dgrad q=(96, 32, 128) K=32x[1-4] " gpurun_out/it/b.err | cut -c1-210
