#!/bin/bash
# C5 frame (10 M photons emitted) under tuning overrides: bash scripts/c5_tuning_sweep.sh "gather_medium=8192" "gather_pair=150" ...
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --photons 2500000 --steps 4 --warmup 1 --tuning "$t" > gpurun_out/c5_g.json 2>/dev/null || exit 1
  echo "[$t] $(python scripts/print_line.py gpurun_out/c5_g.json | cut -c1-40)"
done
