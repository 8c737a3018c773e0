#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel name: python tools/pmc_summary.py <counter_collection.csv>"""
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f"{k:70s} {c:16s} n={len(v):5d} avg={sum(v)/len(v):14.1f}")
