#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/t4.log 2>&1; echo "rc=$?" >> gpurun_out/t4.log; tail -4 gpurun_out/t4.log) || exit 1
grep -q "rc=0" gpurun_out/t4.log || exit 1
bash tools/run_variants.sh gpurun_out/variants4.txt --iters 300 > /dev/null 2>&1; grep fused gpurun_out/variants4.txt
python tools/bench_kernels.py --no-light --cull --iters 200 > gpurun_out/kern4.txt 2>&1; grep "cull" gpurun_out/kern4.txt
UR_CULL_GROUPS_PER_CU=8 python tools/bench_kernels.py --no-light --cull --iters 200 > gpurun_out/kern4_g8.txt 2>&1; grep "cull" gpurun_out/kern4_g8.txt | sed 's/^/G8 /'
