#!/bin/bash
# Counter passes over the photon-mapped frame (C5): k_photon_gather.  On the GPU box: bash scripts/pmc_photon.sh [photons_per_light]
set -e
ppl=${1:-2500000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcp_$name -o $name -- \
      python3 bench.py --photons $ppl --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmcp_$name.log 2>&1
}
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD
run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 scripts/pmc_summary.py gpurun_out/pmcp_summary.txt gpurun_out/pmcp_sq/sq_counter_collection.csv \
    gpurun_out/pmcp_sq2/sq2_counter_collection.csv gpurun_out/pmcp_grbm/grbm_counter_collection.csv
grep k_photon_gather gpurun_out/pmcp_summary.txt
