#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "parity or configs or frame or tail" > gpurun_out/t8.log 2>&1; echo "rc=$?" >> gpurun_out/t8.log; tail -3 gpurun_out/t8.log)
grep -q "rc=0" gpurun_out/t8.log || exit 1
for rep in 1 2 3; do
for rot in 1 0; do
UR_LIGHTING_ROTATE=$rot python tools/bench_kernels.py --gbuffer scene --cache /tmp/urcache --iters 300 --tag "rot$rot " 2>/dev/null | grep fused
done; done
for rot in 1 0; do UR_LIGHTING_ROTATE=$rot python tools/bench_kernels.py --gbuffer scene --width 1920 --height 1080 --iters 1000 --tag "1080p rot$rot " 2>/dev/null | grep fused; done
for rot in 1 0; do UR_LIGHTING_ROTATE=$rot python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench rot$rot', round(d['value']), 'frame_us', round(d['ms_per_step']*1e3,2), 'dispatch_us', round(r['avg_launch_us'],2), 'alone', round(r['alone_on_stream_us'],2))"; done
