#!/bin/bash
# Everything profiles/ holds for a round, from the final sources, in one GPU call:
#   bash scripts/final_round.sh r3
# 1. scripts/profile_round.sh (bench line, kernel-trace stats, timeline, one counter pass per run, derived traffic)
# 2. the derived counters copied into profiles/ of this tree, then the bench line again, so that its roofline object carries them
# 3. the same for C5 (scripts/profile_c5.sh)
set -e
tag=${1:-r3}
cd "$GRAFT_REPO_ROOT"
bash scripts/profile_round.sh $tag > gpurun_out/${tag}_profile_round.log 2>&1 || { tail -20 gpurun_out/${tag}_profile_round.log; exit 1; }
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
bash scripts/profile_c5.sh $tag > gpurun_out/${tag}_profile_c5.log 2>&1 || { tail -20 gpurun_out/${tag}_profile_c5.log; exit 1; }
cp gpurun_out/${tag}_pmc_traffic_c5.json profiles/${tag}_pmc_traffic_c5.json
timeout -k 10 400 python3 bench.py --photons 2500000 --no-cpu-baseline --no-extras > gpurun_out/${tag}_c5_bench.json 2> gpurun_out/${tag}_c5_bench.err || { tail -5 gpurun_out/${tag}_c5_bench.err; exit 1; }
python3 - $tag <<'PY'
import json, sys
t = sys.argv[1]
for name in ("%s_bench.json" % t, "%s_c5_bench.json" % t):
    d = json.loads(open("gpurun_out/" + name).read().strip().splitlines()[-1]); r = d["roofline"]
    print(name, d["value"], d["config"].get("ms_per_frame"), r["avg_launch_ms"], r["frac"], r.get("traffic"), r.get("counters_source"))
PY
