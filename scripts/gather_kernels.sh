#!/bin/bash
# Average duration of the three gather kernels under rocprofv3 --kernel-trace --stats, one run per tuning spec:
#   bash scripts/gather_kernels.sh "gather_fine_cells=8" "gather_fine_cells=16" ...   (RT_HIP_LIBRARY is honoured)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  rm -rf gpurun_out/gk_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gk_prof -o gk -- python3 scripts/gather_sweep.py 2500000 "$spec" > gpurun_out/gk_prof.log 2>&1 || { tail -5 gpurun_out/gk_prof.log; exit 1; }
  python3 - "$spec" <<'PY' | tee -a gpurun_out/gather_kernels.txt
import csv, sys
out = []
for r in csv.DictReader(open("gpurun_out/gk_prof/gk_kernel_stats.csv")):
    n = r["Name"]
    if "photon_gather" in n or "photon_classify" in n or "photon_select" in n or "photon_pass1_group" in n:
        tag = ("group" if "pass1_group" in n else "classify" if "classify" in n else "select_common" if "select_common" in n else "wave<1>" if "wave<true, 1>" in n else "wave<2>" if "wave<true, 2>" in n
               else "wave<0>" if "gather_wave" in n else "lane")
        out.append("%s %.3f ms" % (tag, float(r["AverageNs"]) / 1e6))
import os
print("%-40s %s" % ((os.environ.get("LABEL", "") + " " + (sys.argv[1] or "default")).strip(), "  ".join(sorted(out))))
PY
done
