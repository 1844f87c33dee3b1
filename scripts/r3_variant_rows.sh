#!/bin/bash
# One row per variant LIBRARY of the shadow kernel for profiles/r3_variants/: kernel ms (bench.py's HIP events, unprofiled),
# then one rocprofv3 --pmc pass (instruction counts by class, lanes active, wait share) of the same short command.
#   bash scripts/r3_variant_rows.sh "<bench args>" base variants_tmp/librt_x.so ...      -> gpurun_out/r3_variant_rows.txt
args=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  if [ "$lib" = base ]; then unset RT_HIP_LIBRARY; else export RT_HIP_LIBRARY=$PWD/$lib; fi
  timeout -k 10 200 python3 bench.py $args --steps 5 --warmup 2 --frames-per-step 20 --no-cpu-baseline --no-extras > gpurun_out/var_one.json 2> gpurun_out/var_one.err || { echo "$lib FAILED"; tail -3 gpurun_out/var_one.err; exit 1; }
  rm -rf gpurun_out/var_pmc
  timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH \
      --output-format csv -d gpurun_out/var_pmc -o p -- python3 bench.py $args --steps 2 --warmup 1 --frames-per-step 5 --no-cpu-baseline --no-extras > gpurun_out/var_pmc.log 2>&1 || { echo "$lib pmc FAILED"; exit 1; }
  python3 - "$lib" <<'PY' | tee -a gpurun_out/r3_variant_rows.txt
import csv, json, sys, collections
d = json.loads(open("gpurun_out/var_one.json").read().strip().splitlines()[-1])
r = d["roofline"]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open("gpurun_out/var_pmc/p_counter_collection.csv")):
    if "k_shadow" in row["Kernel_Name"] and "ILb1E" not in row["Kernel_Name"]:
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
m = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
print("%-34s frame %.4f ms  k_shadow %.4f ms | VALU %.3e lanes %.1f%%  SALU %.3e  BRANCH %.3e  VMEM_RD %.3e  LDS %.3e  wait %.1f%%" % (
    sys.argv[1], d["config"]["ms_per_frame"], r["avg_launch_ms"], m.get("SQ_INSTS_VALU", 0),
    100 * m.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * m["SQ_INSTS_VALU"]) if m.get("SQ_INSTS_VALU") else 0, m.get("SQ_INSTS_SALU", 0),
    m.get("SQ_INSTS_BRANCH", 0), m.get("SQ_INSTS_VMEM_RD", 0), m.get("SQ_INSTS_LDS", 0),
    100 * m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0))
PY
done
