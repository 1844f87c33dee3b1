#!/bin/bash
# Profiles of the default bench command for profiles/: kernel-trace stats, then one rocprofv3
# counter pass per run (never combined with the trace domains).  Usage (on the GPU box):
#   bash scripts/profile_round.sh r2      -> gpurun_out/r2_*
set -e
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -o kt -- \
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_kt.err
export PMC_BENCH_ARGS="--steps 2 --warmup 1 --frames-per-step 5 --no-cpu-baseline --no-extras"
bash scripts/pmc_pass.sh fetch FETCH_SIZE
bash scripts/pmc_pass.sh write WRITE_SIZE
bash scripts/pmc_pass.sh tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
bash scripts/pmc_pass.sh sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD
bash scripts/pmc_pass.sh sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM
bash scripts/pmc_pass.sh grbm GRBM_GUI_ACTIVE GRBM_COUNT
bash scripts/pmc_pass.sh tatd TA_TA_BUSY_sum TD_TD_BUSY_sum
bash scripts/pmc_pass.sh tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
python3 scripts/pmc_summary.py $out/${tag}_pmc_summary.txt $out/pmc_fetch/fetch_counter_collection.csv \
    $out/pmc_write/write_counter_collection.csv $out/pmc_tcc/tcc_counter_collection.csv \
    $out/pmc_sq/sq_counter_collection.csv $out/pmc_sq2/sq2_counter_collection.csv $out/pmc_grbm/grbm_counter_collection.csv \
    $out/pmc_tatd/tatd_counter_collection.csv $out/pmc_tcp/tcp_counter_collection.csv
cp $out/${tag}_kt/kt_kernel_stats.csv $out/${tag}_kernel_stats.csv
python3 scripts/timeline.py $out/${tag}_kt/kt_kernel_trace.csv > $out/${tag}_frame_timeline.txt
python3 scripts/kstats.py $out/${tag}_kernel_stats.csv 12
# HBM-side bytes per launch of the dominant kernel and the utilisation figures, for bench.py's roofline object
python3 scripts/pmc_derive.py $out/${tag}_pmc_summary.txt $out/${tag}_pmc_traffic.json "k_shadow<false>" atrium-262k-1080p corrected ploc \
    "python3 bench.py $PMC_BENCH_ARGS (scripts/pmc_pass.sh, one counter pass per run)"
