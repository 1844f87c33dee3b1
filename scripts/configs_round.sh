#!/bin/bash
# The default bench line plus the other measured configurations (profiles/r1_configs/).  On the GPU box.
o=gpurun_out/cfg; mkdir -p $o
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $o/$name.json 2>/dev/null || exit 1; }
timeout -k 10 300 python bench.py > gpurun_out/r1_bench.json 2>gpurun_out/r1_bench.err || exit 1
run atrium-1m-1080p --workload atrium-1m-1080p
run atrium-1m-4k --workload atrium-1m-4k
run cornell-1080p --workload cornell-1080p
run lbvh --accel lbvh
run sah --accel sah
run sah-atrium-1m-1080p --accel sah --workload atrium-1m-1080p
run literal --shadow-mode literal
run inflight-2 --inflight 2
python - <<'PY'
import json, glob
for f in ["gpurun_out/r1_bench.json"] + sorted(glob.glob("gpurun_out/cfg/*.json")):
    d = json.load(open(f)); r = d["roofline"]; c = d["config"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["traffic"], c["bvh_build_ms"], c["bvh_depth"], r["nodes_per_ray"], c["rays_per_frame"])
PY
