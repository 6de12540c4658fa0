#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/t6.log 2>&1; echo "rc=$?" >> gpurun_out/t6.log; tail -4 gpurun_out/t6.log)
grep -q "rc=0" gpurun_out/t6.log || exit 1
python tools/bench_kernels.py --no-light --cull --hzb --iters 200 > gpurun_out/kern6.txt 2>&1; grep "cull\|hzb" gpurun_out/kern6.txt
UR_CULL_STORE=2 python tools/bench_kernels.py --no-light --cull --iters 200 > gpurun_out/kern6_sc1.txt 2>&1; grep "cull" gpurun_out/kern6_sc1.txt | sed 's/^/SC1 /'
python tools/_diag/ride_timing.py > gpurun_out/ride6.txt 2>&1; tail -12 gpurun_out/ride6.txt
