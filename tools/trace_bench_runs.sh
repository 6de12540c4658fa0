#!/bin/bash
# N bench processes, each under rocprofv3 --kernel-trace (no counters); per run the frame time and the inter-kernel gap table
# (tools/trace_gaps.py); the raw trace of a run is kept only when its frame time exceeds the threshold (a "slow" process).
#   bash tools/trace_bench_runs.sh gpurun_out/r2_trace 10 [hzb-launch mode] [slow threshold us]
prefix=${1:-gpurun_out/trace}; n=${2:-5}; mode=${3:-tail-rides}; slow=${4:-93}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 "$n"); do
    rocprofv3 --kernel-trace --output-format csv -d "$R/${prefix}_$i" -- python3 "$R/bench.py" --steps 400 --warmup 5 --no-extras --no-cpu-baseline --hzb-launch "$mode" > "$R/${prefix}_$i.json" 2> "$R/${prefix}_$i.err" || exit 1
    us=$(python3 -c "import json; d=json.loads(open('$R/${prefix}_$i.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,1))")
    echo "== run $i ($mode): $us us/frame"
    python3 "$R/tools/trace_gaps.py" "$R/${prefix}_$i"/*/*kernel_trace.csv | tail -n +2
    if python3 -c "import sys; sys.exit(0 if $us < $slow else 1)"; then rm -rf "$R/${prefix}_$i"; fi
done
