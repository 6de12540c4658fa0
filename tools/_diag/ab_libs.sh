#!/bin/bash
# same-box A/B of the frame over library variants: ab_libs.sh name1 name2 ...   (product = the in-tree build)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
for lib in product "$@"; do
    if [ $lib = product ]; then unset UR_HOTPATH_LIB; else export UR_HOTPATH_LIB=$R/unclerenderer_amd/csrc/_build/variants/libur_$lib.so; fi
    python3 $R/bench.py --no-cpu-baseline --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', round(d['value']), 'frame', round(d['ms_per_step']*1e3,2), 'light', round(r['avg_launch_us'],2), 'alone', round(r['alone_on_stream_us'],2))"
done
done
