#!/bin/bash
# Counter passes over scripts/gather_sweep.py (one rocprofv3 --pmc run per block), per gather kernel:
#   bash scripts/pmc_gather.sh <out-name>     -> gpurun_out/<out-name>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1
rm -rf $out.sq $out.mem $out.grbm
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES \
   --output-format csv -d $out.sq -o p -- python3 scripts/gather_sweep.py 2500000 "" > $out.sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum \
   --output-format csv -d $out.mem -o p -- python3 scripts/gather_sweep.py 2500000 "" > $out.mem.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum \
   --output-format csv -d $out.grbm -o p -- python3 scripts/gather_sweep.py 2500000 "" > $out.grbm.log 2>&1 || exit 1
python3 - $out.sq/p_counter_collection.csv $out.mem/p_counter_collection.csv $out.grbm/p_counter_collection.csv > $out.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "photon_gather" not in k and "photon_select" not in k: continue
        name = ("select" if "select_common" in k else "wave<1>" if "wave<true, 1>" in k else "wave<2>" if "wave<true, 2>" in k
                else "wave<0>" if "gather_wave" in k else "lane")
        a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for name, m in sorted(acc.items()):
    v = {k: x[0] / max(x[1], 1) for k, x in m.items()}
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print("%-8s cycles/XCD %.3e  VALU %.3e (issue %.1f%%, lanes %.1f%%)  SALU %.3e  VMEM_RD %.3e  LDS %.3e  wait %.1f%%  TA %.1f%%  TD %.1f%%  L1 acc %.3e  L1->L2 rd %.3e  L2 hit %.1f%%" % (
        name, cyc, v.get("SQ_INSTS_VALU", 0), 100 * v.get("SQ_INSTS_VALU", 0) * 4 / (1024 * cyc) if cyc else 0,
        100 * v.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * v["SQ_INSTS_VALU"]) if v.get("SQ_INSTS_VALU") else 0,
        v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_LDS", 0),
        100 * v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else 0,
        100 * v.get("TA_TA_BUSY_sum", 0) / (256 * cyc) if cyc else 0, 100 * v.get("TD_TD_BUSY_sum", 0) / (256 * cyc) if cyc else 0,
        v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), v.get("TCP_TCC_READ_REQ_sum", 0),
        100 * v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1)))
PY
cat $out.txt
