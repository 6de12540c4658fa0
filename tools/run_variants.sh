#!/bin/bash
# Time the fused lighting kernel of every diagnostic library under unclerenderer_amd/csrc/_build/variants/ (and the product).
#   bash tools/run_variants.sh gpurun_out/variants.txt [extra bench_kernels.py flags]
out=${1:-gpurun_out/variants.txt}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$(dirname "$R/$out")"
: > "$R/$out"
python3 "$R/tools/bench_kernels.py" --gbuffer scene --cache /tmp/urcache --tag "product " "$@" >> "$R/$out" 2>&1 || exit 1
for lib in "$R"/unclerenderer_amd/csrc/_build/variants/libur_*.so; do
    n=$(basename "$lib" .so); n=${n#libur_}
    UR_HOTPATH_LIB="$lib" python3 "$R/tools/bench_kernels.py" --gbuffer scene --cache /tmp/urcache --tag "$n " "$@" >> "$R/$out" 2>&1 || exit 1
done
python3 "$R/tools/bench_kernels.py" --gbuffer scene --cache /tmp/urcache --tag "product-again " "$@" >> "$R/$out" 2>&1
python3 "$R/tools/bench_kernels.py" --gbuffer scene --cache /tmp/urcache --no-shadows --tag "product-noshadow " "$@" >> "$R/$out" 2>&1
grep "fused" "$R/$out"
