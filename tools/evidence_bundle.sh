#!/bin/bash
# The round's evidence bundle (run on the GPU box): GPU tests, kernel-trace + PMC traffic passes, per-kernel table, bench lines.
#   gpurun --timeout 1100 -- "bash tools/evidence_bundle.sh"; then: python tools/collect_profiles.py r04 r04
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_tests.log; tail -3 gpurun_out/r4_tests.log)
grep -q "rc=0" gpurun_out/r4_tests.log || exit 1
bash tools/profile_round.sh r04 && echo profiled
cd $R && python tools/bench_kernels.py --gbuffer both --hzb --cull --post --iters 300 --cache /tmp/urcache > gpurun_out/r4_kernels.txt 2>&1; cat gpurun_out/r4_kernels.txt | grep -v amdgpu.ids
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_b20.json 2> gpurun_out/r4_b20.err; tail -c 400 gpurun_out/r4_b20.json
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench1000', round(d['value']), 'frame_us', round(d['ms_per_step']*1e3,2), 'dispatch_us', round(r['avg_launch_us'],2), 'frac', round(r['frac'],4), 'alone', round(r['alone_on_stream_us'],2), 'n', r['launches_sampled'])"; done | tee gpurun_out/r4_bench_runs.txt
