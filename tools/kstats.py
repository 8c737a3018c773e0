#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv: python tools/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:96]:96s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot%={100*float(r['TotalDurationNs'])/tot:5.1f}")
