"""print kernels matching a substring from a rocprofv3 kernel_stats csv: name, calls, avg us"""
import csv, glob, os, sys
root, sub = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]:
            print(f"  {r['Name'][:58]:58s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us")
