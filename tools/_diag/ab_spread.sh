#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/tools/bench_kernels.py --gbuffer scene --cache /tmp/urcache --tag warm > /dev/null 2>&1
for rep in 1 2 3; do
    python3 $R/bench.py --no-cpu-baseline --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('product', round(d['value']), 'frame', round(d['ms_per_step']*1e3,2), 'light', round(r['avg_launch_us'],2))"
done
for b in 270 540 1080 2160; do BAND=$b python3 $R/tools/_diag/ride_timing.py auto; done
for b in 270 1080; do UR_RIDE_WALKERS=1 BAND=$b python3 $R/tools/_diag/ride_timing.py one-walker | grep "whole chain"; done
