"""one-line digest of bench.py JSON lines: python benchline.py file.json ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.load(open(f))
    r = d["roofline"]
    print(f, "ms/step", round(d["ms_per_step"], 2), "train", round(d["stage_seconds_per_step"]["step"] * 1e3, 2), "assign",
          round(d["stage_seconds_per_step"]["assign"] * 1e3, 2), "seed", round(d["stage_seconds_per_step"]["seed"] * 1e3, 2),
          "E us", round(r["avg_launch_ms"] * 1e3, 1), "M us", round((r.get("m_step_avg_launch_ms_hip_events") or 0) * 1e3, 1), "frac", round(r["frac"], 3),
          "ingest", {k: round(v["GBps"], 1) for k, v in (d.get("ingest") or {}).items()})
