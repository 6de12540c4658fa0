#!/bin/bash
# same-box A/B of the lighting launch alone over library variants: ab_kernels.sh [reps] name1 name2 ...   (product = the in-tree build)
R=${GRAFT_REPO_ROOT:-/root/repo}
reps=$1; shift
for rep in $(seq 1 $reps); do
for lib in product "$@"; do
    if [ $lib = product ]; then unset UR_HOTPATH_LIB; else export UR_HOTPATH_LIB=$R/unclerenderer_amd/csrc/_build/variants/libur_$lib.so; fi
    python3 $R/tools/bench_kernels.py --gbuffer scene --iters 300 --cache /tmp/urcache --tag "$lib " 2>&1 | grep fused
done
done
