#!/usr/bin/env python3
"""Short form of a bench line: python3 scripts/print_line.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
print(d["value"], "Mrays/s", c["ms_per_frame"], "ms/frame; in flight", c.get("frames_in_flight"), "latency", c.get("frame_latency_ms"),
      "| route:", (c.get("multi_gpu_route") or "-")[:90], "| assembled == whole:", c.get("assembled_frames_equal_whole_frames"),
      "| k_shadow", r["avg_launch_ms"], "frac", r["frac"])
