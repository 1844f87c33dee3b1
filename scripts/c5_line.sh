#!/bin/bash
# The C5 bench line in short form: value, ms/frame, the gather's average launch.
timeout -k 10 400 python3 bench.py --photons ${1:-2500000} --no-cpu-baseline --no-extras > gpurun_out/c5_now.json 2> gpurun_out/c5_now.err || { tail -5 gpurun_out/c5_now.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/c5_now.json").read().strip().splitlines()[-1])
print("C5: value %.2f  ms/frame %.3f  gather launch %.3f ms  (%s)" % (d["value"], d["config"].get("ms_per_frame", 0), d["roofline"]["avg_launch_ms"], d["roofline"]["kernel"][:50]))
PY
