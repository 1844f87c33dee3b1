#!/bin/bash
# One rocprofv3 counter pass over a short bench run (own process, own time limit).
#   scripts/pmc_pass.sh <name> <counter> [<counter> ...]   -> gpurun_out/pmc_<name>/
# Limits per pass on this pool: 8 SQ_, 4 TCC_ (FETCH_SIZE = 3, WRITE_SIZE = 2), 4 TCP_, 2 TA_, 2 TD_, 2 GRBM_.
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o $name -- \
    python3 bench.py ${PMC_BENCH_ARGS:---steps 2 --warmup 1 --frames-per-step 5 --no-cpu-baseline --no-extras} > gpurun_out/pmc_$name.log 2>&1
