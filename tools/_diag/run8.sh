#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/_diag/compare_libs.py 2>/dev/null | sed 's/^/product  /'
for v in identity sched16 sched48; do UR_HOTPATH_LIB=$R/unclerenderer_amd/csrc/_build/variants/libur_$v.so python tools/_diag/compare_libs.py 2>/dev/null | sed "s/^/$v /"; done
bash tools/run_variants.sh gpurun_out/variants8.txt --iters 300 > /dev/null 2>&1; grep fused gpurun_out/variants8.txt
bash tools/run_variants.sh gpurun_out/variants8b.txt --iters 300 > /dev/null 2>&1; grep fused gpurun_out/variants8b.txt
