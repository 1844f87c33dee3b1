#!/bin/bash
# The default bench line plus the other measured configurations (profiles/r2_configs/).  On the GPU box.
o=gpurun_out/cfg; mkdir -p $o
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras "$@" > $o/$name.json 2>/dev/null || exit 1; }
run atrium-1m-1080p --workload atrium-1m-1080p
run atrium-1m-4k --workload atrium-1m-4k
run cornell-1080p --workload cornell-1080p
run lbvh --accel lbvh
run sah --accel sah
run sah-atrium-1m-1080p --accel sah --workload atrium-1m-1080p
run literal --shadow-mode literal
run inflight-2 --inflight 2
run no-pipeline
BENCH_PIPELINE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > $o/no-pipeline.json 2>/dev/null || exit 1
run c5-atrium-262k-10M-photons --photons 2500000 --steps 3 --warmup 1
run c5-atrium-262k-1M-photons --photons 250000 --steps 3 --warmup 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/cfg/*.json")):
    d = json.load(open(f)); r = d["roofline"]; c = d["config"]
    print(f.split("/")[-1], d["value"], c["ms_per_frame"], r["avg_launch_ms"], r["frac"], c["bvh_build_ms"], c["bvh_depth"], r.get("nodes_per_ray"), c["rays_per_frame"])
PY
