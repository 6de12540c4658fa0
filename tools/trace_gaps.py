#!/usr/bin/env python3
"""Inter-kernel gaps of a bench.py run traced with `rocprofv3 --kernel-trace --output-format csv`: per consecutive kernel
pair (cull -> hzb, hzb -> lighting, lighting -> cull) the median / p90 / max gap between one kernel's end and the next one's
start, the kernels' own durations, and the frame period. Used to tell a slow process (frame >= 95 us) from a fast one.

    python tools/trace_gaps.py gpurun_out/trace_*/**/*kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

import numpy as np


def short(name: str) -> str:
    for key, tag in (("lighting_stream_kernel", "light"), ("hzb_reduce", "hzb"), ("hzb_tail", "hzbtail"), ("cull_kernel", "cull"), ("compact", "compact")):
        if key in name:
            return tag
    return name.split("(")[0][-24:]


def analyse(path: Path):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r.get("Queue_Id", 0) or 0)))
    rows.sort()
    # steady state: the last 60 % of the launches (warm-up and extras in front)
    frames = [i for i, r in enumerate(rows) if r[2] == "light"]
    if len(frames) < 50:
        return None
    lo = frames[int(len(frames) * 0.3)]
    hi = frames[int(len(frames) * 0.9)]
    seg = rows[lo:hi]
    gaps, durs = defaultdict(list), defaultdict(list)
    for a, b in zip(seg, seg[1:]):
        gaps[f"{a[2]}->{b[2]}"].append(b[0] - a[1])
        durs[a[2]].append(a[1] - a[0])
    starts = np.array([r[0] for r in seg if r[2] == "light"], dtype=np.float64)
    period = np.diff(starts)
    out = {"file": str(path), "frame_period_us": float(np.median(period)) / 1e3, "queues": sorted({r[3] for r in seg})}
    for k, v in sorted(durs.items()):
        out[f"dur {k}"] = (float(np.median(v)) / 1e3, float(np.percentile(v, 90)) / 1e3)
    for k, v in sorted(gaps.items()):
        if len(v) >= 20:
            out[f"gap {k}"] = (float(np.median(v)) / 1e3, float(np.percentile(v, 90)) / 1e3, float(np.max(v)) / 1e3, len(v))
    return out


def main():
    for p in sys.argv[1:]:
        res = analyse(Path(p))
        if res is None:
            continue
        print(res.pop("file"))
        for k, v in res.items():
            print(f"   {k:28s} {v}")


if __name__ == "__main__":
    main()
