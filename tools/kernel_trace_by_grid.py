#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace per (kernel, grid): one kernel name covers several problem sizes in a run of
tools/bench_kernels.py (Build HZB at 1080p / 4K / 8K, the cull at 25 ... 8 M instances), which `--stats` averages together.

    python tools/kernel_trace_by_grid.py gpurun_out/prof_r03_kernels > profiles/r03_kernels_rocprof.txt
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    path = max(files, key=os.path.getmtime)
    rows = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith(("void at::", "__amd_rocclr")):
            continue  # torch's fills and copies of the harness
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
        rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    print(f"# {os.path.basename(path)}: dispatch durations by kernel and grid (rocprofv3 --kernel-trace; us)")
    print(f"# {'kernel':70s} {'workgroups':>10s} {'threads':>7s} {'vgprs':>5s} {'lds':>7s} {'calls':>6s} {'mean':>8s} {'median':>8s} {'min':>8s}")
    for key in sorted(rows, key=lambda k: (k[0], k[1])):
        v = sorted(rows[key])
        if len(v) < 20:
            continue  # warm-up shapes
        print(f"  {key[0]:70s} {key[1]:10d} {key[2]:7d} {key[3]:5d} {key[4]:7d} {len(v):6d} {sum(v) / len(v):8.2f} {v[len(v) // 2]:8.2f} {v[0]:8.2f}")


if __name__ == "__main__":
    main()
