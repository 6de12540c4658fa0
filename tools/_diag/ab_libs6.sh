#!/bin/bash
# as ab_libs.sh, six interleaved repetitions, then the mean frame time per library
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$(mktemp)
for rep in 1 2 3 4 5 6; do
for lib in product "$@"; do
    if [ $lib = product ]; then unset UR_HOTPATH_LIB; else export UR_HOTPATH_LIB=$R/unclerenderer_amd/csrc/_build/variants/libur_$lib.so; fi
    python3 $R/bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', round(d['value']), 'frame', round(d['ms_per_step']*1e3,2), 'light', round(r['avg_launch_us'],2), 'alone', round(r['alone_on_stream_us'],2))" | tee -a $out
done
done
python3 - $out <<'PY'
import sys, collections
a = collections.defaultdict(list)
for l in open(sys.argv[1]):
    t = l.split()
    a[t[0]].append((float(t[3]), float(t[5])))
for k, v in a.items():
    print("mean", k, "frame %.2f" % (sum(x for x, _ in v) / len(v)), "light %.2f" % (sum(y for _, y in v) / len(v)), "n", len(v))
PY
