#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass (csv):
    python tools/pmc_mfma_util.py <counter_collection.csv> [clock_ghz=2.1] [simds=1024]
SQ_VALU_MFMA_BUSY_CYCLES is the matrix-pipe busy time summed over all SIMDs (it equals passes*4 x the number of MFMA
instructions, e.g. 32 x N for v_mfma_f32_16x16x4_f32), so
    util = busy / (SIMDs * kernel duration * shader clock).
The duration is the dispatch's own start/end timestamp in the same run; the clock is the s_memtime / s_memrealtime ratio
measured inside the kernels (2.1 GHz in the fp32 GEMM, 1.4-1.9 GHz under dense fp16 MFMA)."""
import csv, sys, collections
rows = collections.defaultdict(lambda: {"busy": [], "dur": []})
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        d = rows[r["Kernel_Name"]]
        d["busy"].append(float(r["Counter_Value"]))
        d["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.1
simds = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
out = []
for k, d in rows.items():
    n = len(d["busy"])
    busy, dur = sum(d["busy"]) / n, sum(d["dur"]) / n
    if busy > 0:
        out.append((busy * n, k[:84], n, busy, dur))
print(f"{'kernel':84s} {'calls':>5s} {'avg_us':>8s} {'busy cyc/SIMD':>14s} {'busy cyc/ns':>11s} {'util@%.2fGHz' % ghz:>12s}")
for _, k, n, busy, dur in sorted(out, reverse=True)[:12]:
    per = busy / simds
    print(f"{k:84s} {n:5d} {dur / 1e3:8.1f} {per:14.0f} {per / dur:11.3f} {per / dur / ghz:12.3f}")
