#!/usr/bin/env python3
"""From a pmc_summary.txt (scripts/pmc_summary.py) to the JSON bench.py reads for its roofline object:
HBM-side bytes per launch of one kernel and the utilisation figures the counters give.
    pmc_derive.py summary.txt out.json <kernel name in the summary> <workload> <shadows> <accel> "<command>"
Several kernel names joined by '+' (the photon gather is four kernels, timed and priced as one launch): their per-dispatch
means are added.
"""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
summary, out_path, kernel, workload, shadows, accel, command = sys.argv[1:8]
vals = {}
for line in open(summary):
    m = re.match(r"(.+?)\s+(\S+)\s+dispatches=(\d+)\s+mean=([0-9.e+-]+)", line)   # kernel names may hold a blank (<true, 1>)
    if m and m.group(1) in kernel.split("+"):
        vals[m.group(2)] = vals.get(m.group(2), 0.0) + float(m.group(4))
from bench import kernel_source_hash
SIMDS, TAS, XCDS = 1024, 256, 8
out = {"kernel": kernel, "workload": workload, "shadows": shadows, "accel": accel, "command": command,
       "kernel_source_hash": kernel_source_hash("+photons" in workload),
       "counters_mean_per_launch": vals,
       **({"photons_per_light": int(re.search(r"--photons (\d+)", command).group(1))} if re.search(r"--photons (\d+)", command) else {}),
       "correction": "MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half the bytes of 16 B/lane loads (the node, "
                     "triangle and photon fetches are 16 B/lane) -> x2; WRITE_SIZE as read; unit KB"}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    out["traffic_bytes_per_launch"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
cyc = vals.get("GRBM_GUI_ACTIVE", 0.0) / XCDS      # summed over the XCDs
if cyc > 0:
    out["gpu_cycles_per_launch"] = cyc
    if "SQ_INSTS_VALU" in vals:
        # A SIMD issues a wave64 VALU instruction every 2 cycles when more than one wave feeds it (MI355X_MICROARCH.md; 4 cycles is
        # what ONE wave alone sustains; scripts/probes/pk_rate.hip measures 2.1 cycles at the clock under load).  Rounds 1-2 priced
        # 4 cycles and quoted twice this figure; valu_issue_frac_at_4_cycles keeps that number comparable.
        out["valu_issue_frac"] = round(vals["SQ_INSTS_VALU"] * 2.0 / (SIMDS * cyc), 4)
        out["valu_issue_frac_at_4_cycles"] = round(vals["SQ_INSTS_VALU"] * 4.0 / (SIMDS * cyc), 4)
    if "TA_TA_BUSY_sum" in vals:
        out["ta_busy_frac"] = round(vals["TA_TA_BUSY_sum"] / (TAS * cyc), 4)
    if "TD_TD_BUSY_sum" in vals:
        out["td_busy_frac"] = round(vals["TD_TD_BUSY_sum"] / (TAS * cyc), 4)
if "SQ_THREAD_CYCLES_VALU" in vals and vals.get("SQ_INSTS_VALU"):
    out["lanes_active_frac"] = round(vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_INSTS_VALU"]), 4)
if vals.get("TCC_REQ_sum"):
    out["l2_hit_frac"] = round(vals.get("TCC_HIT_sum", 0.0) / vals["TCC_REQ_sum"], 4)
if vals.get("TCP_TOTAL_CACHE_ACCESSES_sum") and "TCP_TCC_READ_REQ_sum" in vals:
    out["l1_hit_frac"] = round(1.0 - vals["TCP_TCC_READ_REQ_sum"] / vals["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
# Cache-level request bytes per launch (north_star: "achieved L2 GB/s for traversal node fetches against peak"): requests x line size.
# TCC_REQ counts L2 requests of up to one 128-byte line; TCP_TOTAL_CACHE_ACCESSES counts the L1's tag look-ups, 64 bytes each (one
# quarter-wave of a 16 B/lane load).  Both are upper bounds on the bytes that moved (a request may ask for less than a line).
if vals.get("TCC_REQ_sum"):
    out["l2_request_bytes_per_launch"] = vals["TCC_REQ_sum"] * 128.0
if vals.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
    out["l1_request_bytes_per_launch"] = vals["TCP_TOTAL_CACHE_ACCESSES_sum"] * 64.0
if vals.get("SQ_WAVE_CYCLES"):
    out["wait_any_frac"] = round(vals.get("SQ_WAIT_ANY", 0.0) / vals["SQ_WAVE_CYCLES"], 4)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters_mean_per_launch"}))
