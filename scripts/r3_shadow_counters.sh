#!/bin/bash
# Counters of the frame's shadow kernel, ATTRIBUTED PER LAUNCH of the frame (round-3 item: the r2 rows were the mean over the
# three launches of a frame, which trace different ray populations):
#   A = level-0 shadow rays of lights [0, n/2) + the level-0 reflection queue
#   B = level-0 shadow rays of lights [n/2, n) + the level-0 refraction queue
#   C = level-1 shadow rays of all lights
# Four rocprofv3 --pmc passes of the same short bench command (never combined with a trace domain), then one row per class.
#   bash scripts/r3_shadow_counters.sh "<label>" "<bench args>" [kernel name substring]   -> appends to gpurun_out/r3_shadow_rows.txt
set -e
label=$1; args="$2 --steps 2 --warmup 1 --frames-per-step 5 --no-cpu-baseline --no-extras"; kname=${3:-k_shadow}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
timeout -k 10 200 python3 bench.py $2 --steps 5 --warmup 2 --frames-per-step 20 --no-cpu-baseline --no-extras > $out/r3v_bench.json 2> $out/r3v_bench.err
pass() {  # name, counters...
  local name=$1; shift
  rm -rf $out/r3v_$name
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/r3v_$name -o p -- python3 bench.py $args > $out/r3v_$name.log 2>&1
}
pass sq1 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_BUSY_CYCLES
pass ta TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
python3 scripts/r3_shadow_rows.py "$label" "$kname" $out/r3v_bench.json $out/r3v_sq1/p_counter_collection.csv $out/r3v_sq2/p_counter_collection.csv \
    $out/r3v_ta/p_counter_collection.csv $out/r3v_tcc/p_counter_collection.csv >> $out/r3_shadow_rows.txt
tail -5 $out/r3_shadow_rows.txt
