#!/usr/bin/env python3
"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace CSV.
usage: tools/step_timeline.py <dir with *_kernel_trace.csv> [step_from_end]"""
import csv, glob, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
head = "k_step_head" if any("k_step_head" in n for n in names) else "k_pack"   # first kernel of a step
idx = [i for i, n in enumerate(names) if head in n]
starts = [i for k, i in enumerate(idx) if k == 0 or idx[k - 1] != i - 1]
a, b = starts[-back], starts[-back + 1]
t0 = int(rows[a]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    print(f"{(s - t0) / 1e3:8.1f}us gap={(s - prev) / 1e3:6.1f} dur={(e - s) / 1e3:7.1f}  {n[:90]}")
    prev = e; busy += e - s
print("step span us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "busy us", busy / 1e3)
