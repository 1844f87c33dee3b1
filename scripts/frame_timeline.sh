#!/bin/bash
# Kernel timeline of one steady-state frame of the default bench workload: bash scripts/frame_timeline.sh [tag] [bench args]
tag=${1:-tl}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/${tag}_kt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -o kt -- python3 bench.py --steps 3 --warmup 1 --frames-per-step 10 --no-cpu-baseline --no-extras "$@" > $out/${tag}_bench.json 2> $out/${tag}_kt.err || { tail -5 $out/${tag}_kt.err; exit 1; }
python3 scripts/timeline.py $out/${tag}_kt/kt_kernel_trace.csv > $out/${tag}_timeline.txt
cat $out/${tag}_timeline.txt
