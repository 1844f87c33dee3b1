#!/bin/bash
# C5 frame time against the gather's heavy-point threshold (RT_HIP_GATHER_HEAVY).
for t in "$@"; do
  RT_HIP_GATHER_HEAVY=$t timeout -k 10 300 python bench.py --photons 2500000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('heavy', $t, d['ms_per_step'])" || exit 1
done
