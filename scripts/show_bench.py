"""Pretty-print a bench.py JSON line (value, step time, kernel classes with TFLOP/s where known)."""
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    print(f'{d["metric"]}: {d["value"]:.4g} {d["unit"]}  {d["ms_per_step"]:.3f} ms/step  host enqueue {d["host_enqueue_ms_per_step"]:.3f} ms  roofline {d.get("roofline")}')
    tot = 0.0
    for k in d["kernel_classes"]:
        tot += k["ms_per_step"]
        print(f'  {k["tag"]:58s} {k["launches_per_step"]:7.1f} x  {k["ms_per_step"]:8.3f} ms  ({1e3 * k["ms_per_step"] / max(k["launches_per_step"], 1):7.1f} us each)')
    print(f"  sum of listed classes {tot:.3f} ms")
