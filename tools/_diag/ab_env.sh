#!/bin/bash
# same-box A/B of the frame over an environment switch: ab_env.sh VAR v1 v2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
    env $var=$v python3 $R/bench.py --no-cpu-baseline --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$var=$v', round(d['value']), 'frame', round(d['ms_per_step']*1e3,2), 'light', round(r['avg_launch_us'],2), 'alone', round(r['alone_on_stream_us'],2))"
done
done
