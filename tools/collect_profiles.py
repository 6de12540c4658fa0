#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_round.sh into the committed summaries under profiles/.

    python tools/collect_profiles.py r02a r01_stream     # gpurun_out/prof_r02a_* -> profiles/r01_stream_*
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, out = sys.argv[1], sys.argv[2]
import os
newest = lambda pattern: sorted(glob.glob(str(ROOT / pattern), recursive=True), key=os.path.getmtime)[-1:]  # a tag re-run leaves older files behind
stats = newest(f"gpurun_out/prof_{tag}_trace/**/*kernel_stats.csv")
assert stats, "no kernel_stats.csv"
shutil.copy(stats[0], ROOT / f"profiles/{out}_kernel_stats.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write"):
    for f in newest(f"gpurun_out/prof_{tag}_{kind}/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, v in agg.items():
    if "anonymous namespace" not in k:
        continue
    summary[k] = {}
    for c, x in v.items():
        summary[k][f"{c}_KB_avg_per_dispatch"] = round(sum(x) / len(x), 2)
        summary[k][f"dispatches_{c}"] = len(x)
(ROOT / f"profiles/{out}_pmc_fetch_write.json").write_text(json.dumps(summary, indent=1))
light = [k for k in summary if "lighting" in k and "FETCH_SIZE_KB_avg_per_dispatch" in summary[k]]
if light:
    k = max(light, key=lambda n: summary[n]["FETCH_SIZE_KB_avg_per_dispatch"])
    f, w = summary[k]["FETCH_SIZE_KB_avg_per_dispatch"], summary[k]["WRITE_SIZE_KB_avg_per_dispatch"]
    traffic = {
        "lighting_kernel_fused_bytes_per_launch": int(round((2 * f + w) * 1024)),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 300 --warmup 200 "
                  "--no-cpu-baseline --no-extras` (tools/profile_round.sh); per-launch average over the lighting kernel's dispatches; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a wide coalesced read), WRITE_SIZE as is",
        "raw": {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w},
        "kernel": k,
    }
    (ROOT / "profiles/traffic_latest.json").write_text(json.dumps(traffic, indent=1))
    print(json.dumps(traffic, indent=1))
print(open(ROOT / f"profiles/{out}_kernel_stats.csv").read()[:1500])
