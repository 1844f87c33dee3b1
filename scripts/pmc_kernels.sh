#!/bin/bash
# Instruction-class counters per kernel of one short command (one rocprofv3 --pmc pass, no trace domain):
#   bash scripts/pmc_kernels.sh "<kernel name substring>" <command ...>     -> one line per matching kernel
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pk_pmc
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH \
    --output-format csv -d gpurun_out/pk_pmc -o p -- "$@" > gpurun_out/pk_pmc.log 2>&1 || { echo pmc FAILED; tail -3 gpurun_out/pk_pmc.log; exit 1; }
python3 - "$pat" <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for row in csv.DictReader(open("gpurun_out/pk_pmc/p_counter_collection.csv")):
    if sys.argv[1] in row["Kernel_Name"]:
        mm = re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"]); k = mm.group(1) if mm else row["Kernel_Name"][:60]
        a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, c in sorted(acc.items()):
    m = {n: v[0] / max(v[1], 1) for n, v in c.items()}
    print("%-60s n=%-3d VALU %.3e lanes %.1f%%  SALU %.3e  BRANCH %.3e  VMEM_RD %.3e  LDS %.3e  wait %.1f%%" % (
        k, list(c.values())[0][1], m.get("SQ_INSTS_VALU", 0), 100 * m.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * m["SQ_INSTS_VALU"]) if m.get("SQ_INSTS_VALU") else 0,
        m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_BRANCH", 0), m.get("SQ_INSTS_VMEM_RD", 0), m.get("SQ_INSTS_LDS", 0),
        100 * m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0))
PY
