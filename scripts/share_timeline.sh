#!/bin/bash
# Kernel timeline of one virtual rank's share of a strong-scaling frame (scripts/share_probe.py under rocprofv3 --kernel-trace).
#   bash scripts/share_timeline.sh <nranks> <rank> [workload]   -> gpurun_out/share_<n>_<r>_timeline.txt
set -e
n=${1:-8}; r=${2:-0}; wl=${3:-atrium-262k-1080p}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
timeout -k 10 120 python3 scripts/share_probe.py $n $r 50 $wl > $out/share_${n}_${r}.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/share_kt_${n}_${r} -o kt -- python3 scripts/share_probe.py $n $r 20 $wl > $out/share_${n}_${r}_prof.txt 2>&1
python3 scripts/timeline.py $out/share_kt_${n}_${r}/kt_kernel_trace.csv > $out/share_${n}_${r}_timeline.txt
cat $out/share_${n}_${r}.txt $out/share_${n}_${r}_timeline.txt
