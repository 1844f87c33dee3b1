#!/bin/bash
# One line per k_shadow launch-shape variant for profiles/r2_variants/: kernel ms from bench.py's HIP events, then two
# rocprofv3 counter passes of the same command (SQ block; TA/TD + GRBM), never combined with a trace domain.
#   bash scripts/variants_shadow.sh "<label>" "<bench args>"      -> appends to gpurun_out/r2_variants_shadow.txt
set -e
label=$1; shift
args="$* --steps 2 --warmup 1 --frames-per-step 5 --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
timeout -k 10 200 python3 bench.py $* --steps 5 --warmup 2 --frames-per-step 20 --no-cpu-baseline --no-extras > $out/var_bench.json 2> $out/var_bench.err
rm -rf $out/var_sq $out/var_ta
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES \
    --output-format csv -d $out/var_sq -o sq -- python3 bench.py $args > $out/var_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $out/var_ta -o ta -- python3 bench.py $args > $out/var_ta.log 2>&1
python3 - "$label" $out/var_bench.json $out/var_sq/sq_counter_collection.csv $out/var_ta/ta_counter_collection.csv >> $out/r2_variants_shadow.txt <<'PY'
import csv, json, sys, collections
label, bench, *files = sys.argv[1:]
d = json.loads(open(bench).read().strip().splitlines()[-1])
acc = collections.defaultdict(lambda: [0.0, 0])
for path in files:
    for row in csv.DictReader(open(path)):
        if "k_shadow" in row["Kernel_Name"] and "true" not in row["Kernel_Name"].split("k_shadow")[1][:8]:
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
m = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
print("%-44s frame %.4f ms  k_shadow %.4f ms  | VALU %.3e (issue %.1f%%, lanes %.1f%%)  SALU %.3e  VMEM_RD %.3e  LDS %.3e  wait %.1f%%  TA %.1f%%  TD %.1f%%" % (
    label, d["config"]["ms_per_frame"], d["roofline"]["avg_launch_ms"], m.get("SQ_INSTS_VALU", 0),
    100 * m.get("SQ_INSTS_VALU", 0) * 4 / (1024 * cyc) if cyc else 0,
    100 * m.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * m["SQ_INSTS_VALU"]) if m.get("SQ_INSTS_VALU") else 0,
    m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_VMEM_RD", 0), m.get("SQ_INSTS_LDS", 0),
    100 * m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0,
    100 * m.get("TA_TA_BUSY_sum", 0) / (256 * cyc) if cyc else 0, 100 * m.get("TD_TD_BUSY_sum", 0) / (256 * cyc) if cyc else 0))
PY
tail -1 $out/r2_variants_shadow.txt
