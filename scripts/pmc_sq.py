"""Aggregate a rocprofv3 --pmc counter_collection csv per kernel: mean counter value per dispatch.
usage: python scripts/pmc_sq.py <dir> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict

root, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            a = acc[k[:60]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print(f"  {c:34s} {s / n:16.0f}  (n={n})")
