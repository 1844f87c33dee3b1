#!/bin/bash
# Counter passes of the C5 bench command (photon-mapped frame, 10 M photons emitted) for profiles/: what bench.py --photons
# reads back for its roofline object.  One rocprofv3 --pmc run per block.  Usage (on the GPU box):
#   bash scripts/profile_c5.sh r2      -> gpurun_out/r2_c5_*
set -e
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
export PMC_BENCH_ARGS="--photons 2500000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
timeout -k 10 300 python3 bench.py --photons 2500000 --no-cpu-baseline --no-extras > $out/${tag}_c5_bench.json 2> $out/${tag}_c5_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c5_kt -o kt -- \
    python3 bench.py $PMC_BENCH_ARGS --steps 3 > $out/${tag}_c5_bench_under_rocprof.json 2> $out/${tag}_c5_kt.err
cp $out/${tag}_c5_kt/kt_kernel_stats.csv $out/${tag}_c5_kernel_stats.csv
bash scripts/pmc_pass.sh c5fetch FETCH_SIZE
bash scripts/pmc_pass.sh c5write WRITE_SIZE
bash scripts/pmc_pass.sh c5tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
bash scripts/pmc_pass.sh c5sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD
bash scripts/pmc_pass.sh c5grbm GRBM_GUI_ACTIVE GRBM_COUNT
bash scripts/pmc_pass.sh c5tatd TA_TA_BUSY_sum TD_TD_BUSY_sum
bash scripts/pmc_pass.sh c5tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
python3 scripts/pmc_summary.py $out/${tag}_c5_pmc_summary.txt $out/pmc_c5fetch/c5fetch_counter_collection.csv \
    $out/pmc_c5write/c5write_counter_collection.csv $out/pmc_c5tcc/c5tcc_counter_collection.csv $out/pmc_c5sq/c5sq_counter_collection.csv \
    $out/pmc_c5grbm/c5grbm_counter_collection.csv $out/pmc_c5tatd/c5tatd_counter_collection.csv $out/pmc_c5tcp/c5tcp_counter_collection.csv
python3 scripts/kstats.py $out/${tag}_c5_kernel_stats.csv 8
python3 scripts/pmc_derive.py $out/${tag}_c5_pmc_summary.txt $out/${tag}_pmc_traffic_c5.json \
    "k_photon_gather+k_photon_gather_wave<true, 1>+k_photon_gather_wave<true, 2>+k_photon_select_common" atrium-262k-1080p+photons corrected ploc \
    "python3 bench.py $PMC_BENCH_ARGS (scripts/pmc_pass.sh, one counter pass per run; the four gather kernels of a launch added)"
