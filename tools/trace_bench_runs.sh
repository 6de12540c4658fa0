#!/bin/bash
# N bench processes, each under rocprofv3 --kernel-trace (no counters), traces kept per run under gpurun_out/<prefix>_<i>/
#   bash tools/trace_bench_runs.sh gpurun_out/r2_trace 5
prefix=${1:-gpurun_out/trace}; n=${2:-5}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 "$n"); do
    rocprofv3 --kernel-trace --output-format csv -d "$R/${prefix}_$i" -- python3 "$R/bench.py" --steps 400 --warmup 5 --no-extras --no-cpu-baseline > "$R/${prefix}_$i.json" 2> "$R/${prefix}_$i.err" || exit 1
    python3 -c "import json,sys; d=json.loads(open('$R/${prefix}_$i.json').read().strip().splitlines()[-1]); print('run $i', round(d['ms_per_step']*1e3,1), 'us/frame, lighting', round(d['roofline']['avg_launch_us'],1))"
done
