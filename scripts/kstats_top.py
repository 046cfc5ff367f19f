"""top kernels of a rocprofv3 --kernel-trace --stats run: name, calls, total ms, avg us, share"""
import csv, glob, os, sys
root = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 30; div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{f}: {len(rows)} kernels, total {tot / 1e6:.3f} ms ({tot / 1e6 / div:.3f} ms per step over {div:g} steps), {sum(int(r['Calls']) for r in rows) / div:.0f} launches per step")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        print(f"  {r['Name'][:70]:70s} {int(r['Calls']) / div:8.1f} x {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['TotalDurationNs']) / 1e6 / div:8.3f} ms  {100 * float(r['TotalDurationNs']) / tot:5.1f} %")
