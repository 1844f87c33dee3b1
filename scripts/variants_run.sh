#!/bin/bash
# bench.py (short form, no extras) once per variant library: bash scripts/variants_run.sh "<bench args>" lib1.so lib2.so ...
# "base" stands for the in-tree library.  One line per library into gpurun_out/variants_run.txt.
args=$1; shift
for lib in "$@"; do
  if [ "$lib" = base ]; then unset RT_HIP_LIBRARY; else export RT_HIP_LIBRARY=$PWD/$lib; fi
  timeout -k 10 200 python3 bench.py $args --steps 5 --warmup 2 --frames-per-step 20 --no-cpu-baseline --no-extras > gpurun_out/var_one.json 2> gpurun_out/var_one.err || { echo "$lib FAILED"; tail -3 gpurun_out/var_one.err; exit 1; }
  python3 - "$lib" <<'PY' | tee -a gpurun_out/variants_run.txt
import json, sys
d = json.loads(open("gpurun_out/var_one.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("%-36s frame %.4f ms  %s %.4f ms  value %.1f" % (sys.argv[1], d["config"]["ms_per_frame"], r["kernel"].split()[0], r["avg_launch_ms"], d["value"]))
PY
done
