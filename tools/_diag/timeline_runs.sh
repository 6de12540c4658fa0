#!/bin/bash
# N default bench processes with the GPU-side timeline on: frame time + span/gap summary per process
R=${GRAFT_REPO_ROOT:-/root/repo}
n=${1:-10}; shift
for i in $(seq 1 $n); do
  python3 $R/bench.py --no-cpu-baseline --no-extras --timeline "$@" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('timeline') or {}
print('run $i', round(d['value']), 'frame', round(d['ms_per_step']*1e3,2), 'host', round(d['host_submit_ms_per_step']*1e3,1), '| light span', t.get('lighting_span_us'), 'cull span', t.get('other_span_us'), 'gap->light', t.get('gap_before_lighting_us'), 'gap->cull', t.get('gap_before_other_us'), '| gaps', t.get('largest_gaps'), 'host', t.get('largest_host_pauses'), t.get('host_submit_us_median'))"
done
