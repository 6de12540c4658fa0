#!/bin/bash
# Round profile bundle (run on the GPU box): kernel-trace stats and the HBM-traffic PMC passes of `bench.py`.
#   bash tools/profile_round.sh r02          -> gpurun_out/prof_r02_{trace,fetch,write}
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 300 --warmup 200 --no-cpu-baseline --no-extras"
# the kernel-trace pass runs 4000 timed frames: the profiler averages over EVERY dispatch, the ~400 of the clock ramp (80+ us each) included,
# and with 300 timed frames those were a fifth of the sample (73.0 us against 71.4 in the steady part of the same trace)
BT="python3 $R/bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_${tag}_trace" -- $BT > "$R/gpurun_out/prof_${tag}_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/prof_${tag}_fetch" -- $B > "$R/gpurun_out/prof_${tag}_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/prof_${tag}_write" -- $B > "$R/gpurun_out/prof_${tag}_write.log" 2>&1
