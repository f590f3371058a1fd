#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (averages over dispatches)."""
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    if n.startswith("void "):
        n = n[5:]
    if not n.startswith("k_"):          # every kernel of libcinema_hip.so is named k_*; torch / rocprim kernels are not
        continue
    agg[n[:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print(f"{n:64s} " + " ".join(f"{k}={v:.4g}" for k, v in sorted(m.items())))
