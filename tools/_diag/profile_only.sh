#!/bin/bash
# the rocprofv3 passes of the evidence bundle plus the bench lines of the same box (no tests, no per-kernel table)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_round.sh r03 && echo profiled
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/b20_7.json 2> gpurun_out/b20_7.err; tail -c 300 gpurun_out/b20_7.json
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench1000', round(d['value']), 'frame_us', round(d['ms_per_step']*1e3,2), 'dispatch_us', round(r['avg_launch_us'],2), 'frac', round(r['frac'],4), 'alone', round(r['alone_on_stream_us'],2), 'n', r['launches_sampled'])"; done | tee gpurun_out/bench_runs7.txt
