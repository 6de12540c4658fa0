#!/bin/bash
# round-3 batch: parity for the single-launch cull, lighting variants, cull timings, HZB per-launch profile, short bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/t3.log 2>&1; echo "rc=$?" >> gpurun_out/t3.log; tail -4 gpurun_out/t3.log) || exit 1
bash tools/run_variants.sh gpurun_out/variants3.txt --iters 300 > /dev/null 2>&1; grep fused gpurun_out/variants3.txt
python tools/bench_kernels.py --no-light --cull --hzb --iters 200 > gpurun_out/kern3.txt 2>&1; grep "cull\|hzb" gpurun_out/kern3.txt
UR_CULL_NT=1 python tools/bench_kernels.py --no-light --cull --iters 200 > gpurun_out/kern3_nt.txt 2>&1; grep "cull" gpurun_out/kern3_nt.txt | sed 's/^/NT /'
UR_CULL_GROUPS_PER_CU=8 python tools/bench_kernels.py --no-light --cull --iters 200 > gpurun_out/kern3_g8.txt 2>&1; grep "cull" gpurun_out/kern3_g8.txt | sed 's/^/G8 /'
UR_CULL_GROUPS_PER_CU=2 python tools/bench_kernels.py --no-light --cull --iters 200 > gpurun_out/kern3_g2.txt 2>&1; grep "cull" gpurun_out/kern3_g2.txt | sed 's/^/G2 /'
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_hzb3 -o hzb -- python3 $R/tools/bench_kernels.py --no-light --hzb --iters 100 > $R/gpurun_out/prof_hzb3.log 2>&1
cd $R && python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/b20_3.json 2> gpurun_out/b20_3.err; python -c "
import json; d=json.loads(open('gpurun_out/b20_3.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench20 frame_us', round(d['ms_per_step']*1e3,2), 'value', round(d['value']), 'dispatch', round(r['avg_launch_us'],2), 'frac', round(r['frac'],4), 'n', r['launches_sampled'], 'gc_on_frame', round(d['with_python_gc_on']['ms_per_step']*1e3,2))"
