#!/bin/bash
# Profiles of the default bench command for profiles/: kernel-trace stats, then one rocprofv3
# counter pass per run (never combined with the trace domains).  Usage (on the GPU box):
#   bash scripts/profile_round.sh r1      -> gpurun_out/r1_*
set -e
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -o kt -- \
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_kt.err
bash scripts/pmc_pass.sh fetch FETCH_SIZE
bash scripts/pmc_pass.sh write WRITE_SIZE
bash scripts/pmc_pass.sh tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
bash scripts/pmc_pass.sh sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD
bash scripts/pmc_pass.sh sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM
bash scripts/pmc_pass.sh grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 scripts/pmc_summary.py $out/${tag}_pmc_summary.txt $out/pmc_fetch/fetch_counter_collection.csv \
    $out/pmc_write/write_counter_collection.csv $out/pmc_tcc/tcc_counter_collection.csv \
    $out/pmc_sq/sq_counter_collection.csv $out/pmc_sq2/sq2_counter_collection.csv $out/pmc_grbm/grbm_counter_collection.csv
cp $out/${tag}_kt/kt_kernel_stats.csv $out/${tag}_kernel_stats.csv
python3 scripts/timeline.py $out/${tag}_kt/kt_kernel_trace.csv > $out/${tag}_frame_timeline.txt
python3 scripts/kstats.py $out/${tag}_kernel_stats.csv 12
# HBM-side bytes per launch of the dominant kernel, for bench.py's roofline.traffic
python3 - "$out/${tag}_pmc_summary.txt" "$out/${tag}_pmc_traffic.json" <<'PY'
import json, re, sys
vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"(k_shadow<false>)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches=\d+\s+mean=([0-9.e+]+)", line)
    if m:
        vals[m.group(2)] = float(m.group(3))
out = {"kernel": "k_shadow<false>",
       "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (scripts/pmc_pass.sh, one counter pass per run)",
       "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
       "correction": "MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half the bytes of 16 B/lane loads (the node and triangle fetches are 16 B/lane) -> x2; WRITE_SIZE as read; unit KB",
       "traffic_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out["traffic_bytes_per_launch"])
PY
