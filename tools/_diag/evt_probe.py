"""What the three timing forms report for the same fused Lighting launch inside the frame, and what each costs the frame
(bench.py picks the one that matches rocprofv3's kernel trace and costs the least)."""
import json, subprocess, sys
from pathlib import Path
R = Path(__file__).resolve().parents[2]
for every in ("1", "2", "4", "0"):
    pr = subprocess.run([sys.executable, str(R / "bench.py"), "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"] + (["--no-light-events"] if every == "0" else ["--light-every", every]),
                        capture_output=True, text=True)
    if not pr.stdout.strip():
        print("every", every, "FAILED", pr.stderr[-1500:])
        continue
    out = pr.stdout.strip().splitlines()[-1]
    d = json.loads(out); r = d["roofline"]
    print("every", every, "frame_us", round(d["ms_per_step"] * 1e3, 2), "dispatch_us", round(r["avg_launch_us"], 2), "median", round(r["median_launch_us"], 2), "n", r["launches_sampled"],
          "bracket", round(r["event_bracket_us"], 2), "record", round(r["event_record_us"], 2), "gc_on_frame_us", round(d.get("with_python_gc_on", {}).get("ms_per_step", 0) * 1e3, 2))
