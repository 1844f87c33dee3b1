#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pr_prof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pr_prof -o pr -- python3 scripts/process_rays_timeline.py 20 > gpurun_out/pr_prof.log 2>&1 || { tail -5 gpurun_out/pr_prof.log; exit 1; }
tail -1 gpurun_out/pr_prof.log
python3 - <<'PY'
import csv
rows = sorted(csv.DictReader(open("gpurun_out/pr_prof/pr_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
# the last call: kernels after the last k_raygen-like first kernel
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "gbuffer_rays" in n or "raygen" in n or "reflect" in n.lower()) if any(("gbuffer_rays" in n or "raygen" in n or "reflect" in n.lower()) for n in names) else len(rows) - 12
t0 = int(rows[last]["Start_Timestamp"])
prev_end = t0
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s start %8.1f dur %8.1f gap %6.1f" % (r["Kernel_Name"][:60], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
PY
