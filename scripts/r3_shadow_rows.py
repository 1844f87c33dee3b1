#!/usr/bin/env python3
"""One row per launch class (A, B, C, mean) of the frame's shadow kernel from rocprofv3 counter CSVs
(scripts/r3_shadow_counters.sh).  A frame launches the kernel three times, in order A, B, C."""
import collections
import csv
import json
import sys

label, kname, bench, *files = sys.argv[1:]
d = json.loads(open(bench).read().strip().splitlines()[-1])
acc = {c: collections.defaultdict(lambda: [0.0, 0]) for c in "ABC"}
for path in files:
    rows = [r for r in csv.DictReader(open(path)) if kname in r["Kernel_Name"] and "ILb1E" not in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    cls = {i: "ABC"[k % 3] for k, i in enumerate(ids)}
    for r in rows:
        a = acc[cls[int(r["Dispatch_Id"])]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
print("%s   frame %.4f ms   kernel %.4f ms/launch (mean of A, B, C; HIP events, unprofiled run)" % (label, d["config"]["ms_per_frame"], d["roofline"]["avg_launch_ms"]))
means = {}
for c in "ABC":
    means[c] = {k: v[0] / max(v[1], 1) for k, v in acc[c].items()}
means["mean"] = {k: sum(means[c].get(k, 0.0) for c in "ABC") / 3 for k in means["A"]}
for c in ("A", "B", "C", "mean"):
    m = means[c]
    g = lambda k: m.get(k, 0.0)
    cyc = g("GRBM_GUI_ACTIVE") / 8          # cycles per XCD for the launch
    simd_cycles = cyc * 1024                # SIMD-cycles of the chip
    wc = g("SQ_WAVE_CYCLES")
    print("  %-4s cycles/XCD %7.0fk | VALU %.3e lanes %.1f%% | SALU %.3e | VMEM_RD %.3e LDS %.3e SMEM %.3e BRANCH %.3e | waves %6.0f | "
          "busy of wave-cycles: VALU %.1f%% SCA %.1f%% VMEM %.1f%% LDS %.1f%% any %.1f%% | wait any %.1f%% wait-inst %.1f%% | TA %.1f%% TD %.1f%% | "
          "L1 req %.3e L1->L2 rd %.3e (L1 hit %.1f%%) L2 req %.3e hit %.1f%% | ifetch %.3e" % (
              c, cyc / 1e3, g("SQ_INSTS_VALU"), 100 * g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_INSTS_VALU")) if g("SQ_INSTS_VALU") else 0,
              g("SQ_INSTS_SALU"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_LDS"), g("SQ_INSTS_SMEM"), g("SQ_INSTS_BRANCH"), g("SQ_WAVES"),
              100 * g("SQ_ACTIVE_INST_VALU") / wc if wc else 0, 100 * g("SQ_ACTIVE_INST_SCA") / wc if wc else 0,
              100 * g("SQ_ACTIVE_INST_VMEM") / wc if wc else 0, 100 * g("SQ_ACTIVE_INST_LDS") / wc if wc else 0,
              100 * g("SQ_ACTIVE_INST_ANY") / wc if wc else 0, 100 * g("SQ_WAIT_ANY") / wc if wc else 0, 100 * g("SQ_WAIT_INST_ANY") / wc if wc else 0,
              100 * g("TA_TA_BUSY_sum") / (256 * cyc) if cyc else 0, 100 * g("TD_TD_BUSY_sum") / (256 * cyc) if cyc else 0,
              g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("TCP_TCC_READ_REQ_sum"),
              100 * (1 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")) if g("TCP_TOTAL_CACHE_ACCESSES_sum") else 0,
              g("TCC_REQ_sum"), 100 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") + g("TCC_MISS_sum") else 0,
              g("SQ_IFETCH")))
