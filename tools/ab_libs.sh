#!/bin/bash
# Same-box A/B of library builds (the in-tree product against unclerenderer_amd/csrc/_build/variants/libur_<name>.so), one process per
# library and repetition, interleaved:   bash tools/ab_libs.sh OUT.txt REPS name1 [name2 ...] [-- extra ab_options.py flags]
out=$1; reps=$2; shift 2
names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ "$1" = "--" ] && shift
R=${GRAFT_REPO_ROOT:-/root/repo}
: > "$R/$out"
for rep in $(seq 1 "$reps"); do
  for lib in product "${names[@]}"; do
    if [ "$lib" = product ]; then unset UR_HOTPATH_LIB; else export UR_HOTPATH_LIB=$R/unclerenderer_amd/csrc/_build/variants/libur_$lib.so; fi
    echo "## $lib rep $rep" >> "$R/$out"
    timeout -k 10 300 python3 "$R/tools/ab_options.py" "$lib:" --rounds 3 --iters 1500 --settle 2500 "$@" 2>&1 | grep -v amdgpu.ids | tail -4 >> "$R/$out" || exit 1
  done
done
grep -E "^(product|${names[0]})" "$R/$out" | sort
