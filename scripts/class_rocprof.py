"""profiles/rNN_class_rocprof.json from a rocprofv3 kernel-stats csv of `bench.py`: the average launch duration of each kernel-timer CLASS
(the tags bench.py's roofline object names), so that the bench line can carry the profiler's figure next to its own event-bracketed one.
usage: python scripts/class_rocprof.py <kernel_stats.csv> <out.json> [note]"""
import csv
import json
import re
import sys

CLASSES = {
    "conv_igemm_fprop_dgrad": r"^(void )?(lamp::)?ig_conv8[bcd]_kernel<",
    "conv_wgrad_igemm": r"^(void )?(lamp::)?ig_wgrad8(h|v2)_kernel<",
    "conv_narrow_fprop_dgrad": r"^(void )?(lamp::)?ncv_fwd2?_kernel<",
    "conv_wgrad_narrow": r"^(void )?(lamp::)?ncv_wgrad2?_kernel<",
    "conv_wgrad_reduce": r"^(void )?(lamp::)?wgrad_reduce_many_kernel",
    "gemm_bf16_pp2": r"^(void )?(lamp::)?gemm_bf16_pp2_kernel<",
}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = list(csv.DictReader(open(src)))
    out = {}
    for tag, pat in CLASSES.items():
        calls = ns = 0
        names = []
        for r in rows:
            if re.search(pat, r["Name"]):
                calls += int(r["Calls"]); ns += float(r["TotalDurationNs"]); names.append(r["Name"][:60])
        if calls:
            out[tag] = {"avg_us": ns / calls / 1e3, "calls": calls, "total_us": ns / 1e3, "kernels": names}
    json.dump({"source": src, "note": note, "classes": out}, open(dst, "w"), indent=1)
    for k, v in out.items():
        print(f"{k:28s} calls {v['calls']:6d} avg {v['avg_us']:8.2f} us")


if __name__ == "__main__":
    main()
